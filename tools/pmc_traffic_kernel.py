"""HBM-side traffic of the level-0 projection-fused launch as bench.py issues it (head-pair kernel, y in query-fragment order),
keyed to the kernel's SOURCES: two rocprofv3 passes (--kernel-trace --pmc FETCH_SIZE, then --pmc WRITE_SIZE; never combined with
trace domains) over tools/proj_bench.py, folded into gpurun_out/xattn_fwd_hbm_traffic.json = profiles/xattn_fwd_hbm_traffic.json
+ the entry by_kernel["proj_N4096_C320_I<imgs>"] with `source_sha` (bench.py::source_sha). bench.py reports `roofline.traffic`
only when that hash equals the hash of the sources it runs. Run on the GPU box from the repo root:
    python tools/pmc_traffic_kernel.py [--imgs 32] [--dtype fp16]      then copy the JSON to profiles/."""
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
ap.add_argument("--imgs", type=int, default=32)
ap.add_argument("--dtype", default="fp16")
a = ap.parse_args()
N, C, K, M = 4096, 320, 2, 77
raw = {}
env = dict(os.environ, TMPDIR="/tmp")
for cnt in ("FETCH_SIZE", "WRITE_SIZE"):
    d = "/tmp/pmck_%s" % cnt
    subprocess.run(["rm", "-rf", d])
    r = subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", cnt, "-d", d, "-o", "k", "--", sys.executable, os.path.join(ROOT, "tools", "proj_bench.py"),
                        "--only", "pairqo", "--iters", "10", "--rounds", "1", "--imgs", str(a.imgs), "--dtype", a.dtype], cwd="/tmp", env=env,
                       capture_output=True, text=True)
    db = [os.path.join(dp, f) for dp, _, fs in os.walk(d) for f in fs if f.endswith("_results.db")]
    assert db, r.stderr[-2000:]
    rows = [x for x in pmc_stats(db[0]) if "proj_p3" in x[0] and x[2] == cnt]
    assert len(rows) == 1, rows
    raw[cnt] = round(rows[0][4], 1)
    kernel = rows[0][0][:110]
import bench  # noqa: E402  (source_sha)
alg = a.imgs * (8 * N * C + 4 * (K + 2) * M * C + K * N) + 2 * C * C
ent = {"kernel": kernel + " (to_q inside, a head pair per workgroup, y in query-fragment order), N=%d C=%d K=%d, %d images per launch" % (N, C, K, a.imgs),
       "raw_KiB": raw, "bytes_per_launch": (2 * raw["FETCH_SIZE"] + raw["WRITE_SIZE"]) * 1024, "algorithmic_bytes": alg, "dtype": a.dtype,
       "source_sha": bench.source_sha(),
       "how": "tools/pmc_traffic_kernel.py: separate rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE passes over tools/proj_bench.py --only pairqo "
              "(KiB per dispatch, average of 16 launches); FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md); round 5"}
ent["ratio_to_algorithmic"] = round(ent["bytes_per_launch"] / alg, 4)
out = os.path.join(ROOT, "gpurun_out", "xattn_fwd_hbm_traffic.json")          # several runs of one GPU call accumulate here
src = out if os.path.exists(out) else os.path.join(ROOT, "profiles", "xattn_fwd_hbm_traffic.json")
doc = json.load(open(src)) if os.path.exists(src) else {}
doc.setdefault("by_kernel", {})["proj_N%d_C%d_I%d" % (N, C, a.imgs) + ("" if a.dtype == "fp16" else "_" + a.dtype)] = ent
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(doc, open(os.path.join(ROOT, "gpurun_out", "xattn_fwd_hbm_traffic.json"), "w"), indent=1)
print(json.dumps(ent, indent=1))
