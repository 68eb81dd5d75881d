"""HBM-side traffic of sta_xattn_bwd's kernel at the four SD-v1 level shapes as the tracked epochs launch it (16 images per launch),
keyed to the kernel's SOURCES: per level two rocprofv3 passes (--kernel-trace --pmc FETCH_SIZE, then --pmc WRITE_SIZE; never
combined with trace domains) over tools/kernel_bench.py --bwd, written to gpurun_out/xattn_bwd_hbm_traffic.json (copy to profiles/).
bench.py reports `roofline_bwd.traffic` only when the entry's `source_sha` equals the hash of the sources it runs.
    python tools/pmc_traffic_bwd.py [--imgs 16] [--dtype fp16]"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from rocpd_stats import pmc_stats  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--imgs", type=int, default=16)
ap.add_argument("--dtype", default="fp16")
a = ap.parse_args()
K, M = 2, 77
LEVELS = [(4096, 320), (1024, 640), (256, 1280), (64, 1280)]
import bench  # noqa: E402  (source_sha)
sha = bench.source_sha(("sta_xattn_bwd.hip", "sta_xattn_dev.h"))
env = dict(os.environ, TMPDIR="/tmp")
doc = {"by_kernel": {}}
for L, (N, C) in enumerate(LEVELS):
    raw, kernel = {}, None
    for cnt in ("FETCH_SIZE", "WRITE_SIZE"):
        d = "/tmp/pmcb_%d_%s" % (L, cnt)
        subprocess.run(["rm", "-rf", d])
        r = subprocess.run(["timeout", "240", "rocprofv3", "--kernel-trace", "--pmc", cnt, "-d", d, "-o", "k", "--", sys.executable,
                            os.path.join(ROOT, "tools", "kernel_bench.py"), "--bwd", "--iters", "10", "--imgs", str(a.imgs), "--level", str(L), "--dtype", a.dtype],
                           cwd="/tmp", env=env, capture_output=True, text=True)
        db = [os.path.join(dp, f) for dp, _, fs in os.walk(d) for f in fs if f.endswith("_results.db")]
        assert db, r.stderr[-2000:]
        rows = [x for x in pmc_stats(db[0]) if "xattn_bwd_res" in x[0] and x[2] == cnt]
        assert len(rows) == 1, rows
        raw[cnt] = round(rows[0][4], 1)
        kernel = rows[0][0][:110]
    alg = a.imgs * (12 * N * C + 4 * (K + 2) * M * C + K * N)
    ent = {"kernel": kernel + ", N=%d C=%d K=%d, %d images per launch" % (N, C, K, a.imgs), "raw_KiB": raw,
           "bytes_per_launch": (2 * raw["FETCH_SIZE"] + raw["WRITE_SIZE"]) * 1024, "algorithmic_bytes": alg, "dtype": a.dtype, "source_sha": sha,
           "how": "tools/pmc_traffic_bwd.py: separate rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE passes over tools/kernel_bench.py --bwd "
                  "(KiB per dispatch, average of 20 launches); FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md); round 6"}
    ent["ratio_to_algorithmic"] = round(ent["bytes_per_launch"] / alg, 4)
    doc["by_kernel"]["bwd_N%d_C%d_I%d" % (N, C, a.imgs) + ("" if a.dtype == "fp16" else "_" + a.dtype)] = ent
    print(json.dumps(ent, indent=1), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
out = os.path.join(ROOT, "gpurun_out", "xattn_bwd_hbm_traffic.json")
if os.path.exists(out):
    old = json.load(open(out))
    old.setdefault("by_kernel", {}).update(doc["by_kernel"])
    doc = old
json.dump(doc, open(out, "w"), indent=1)
