"""Fold the rocprofv3 PMC databases written by tools/pmc_traffic.sh into profiles/xattn_fwd_hbm_traffic.json."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from rocpd_stats import pmc_stats  # noqa: E402

LEVELS = [(4096, 320, 5), (1024, 640, 5), (256, 1280, 5), (64, 1280, 1)]   # N, C, launches per UNet call
K, M = 2, 77
root = sys.argv[1]
out = {"by_images_per_launch": {}}
for I in (1, 8, 16):
    rec = {"per_level_bytes": {}, "algorithmic_bytes": {}, "launches_per_unet_call": {}, "raw_KiB": {}, "kernel": {}}
    tot = n = 0
    for L, (N, C, cnt) in enumerate(LEVELS):
        key = "N%d_C%d" % (N, C)
        raw = {}
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            rows = [r for r in pmc_stats(os.path.join(root, "%d_%d_%s" % (I, L, counter), "k_results.db"))
                    if "xattn_fwd" in r[0] and r[2] == counter]
            assert len(rows) == 1, rows
            raw[counter] = round(rows[0][4], 1)
            rec["kernel"][key] = rows[0][0][:90]
        rec["raw_KiB"][key] = raw
        rec["per_level_bytes"][key] = (2 * raw["FETCH_SIZE"] + raw["WRITE_SIZE"]) * 1024
        rec["algorithmic_bytes"][key] = I * (8 * N * C + 4 * (K + 2) * M * C + K * N)
        rec["launches_per_unet_call"][key] = cnt
        tot += cnt * rec["per_level_bytes"][key]
        n += cnt
    rec["bytes_per_launch"] = tot / n
    rec["how"] = ("tools/pmc_traffic.sh: separate rocprofv3 passes (--kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE) over "
                  "tools/kernel_bench.py --iters 20 --imgs %d --level L; counters are KiB per dispatch (average); traffic = "
                  "(2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE counts 128-B requests as 64 B, MI355X_MICROARCH.md HBM "
                  "section); Infinity-Cache hits are included in the counters" % I)
    out["by_images_per_launch"][str(I)] = rec
out["bytes_per_launch"] = out["by_images_per_launch"]["1"]["bytes_per_launch"]
print(json.dumps(out, indent=1))
