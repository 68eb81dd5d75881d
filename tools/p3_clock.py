"""Why does the fp16 head-pair launch run a few % behind the bf16 one on an instruction stream that is the same line for line?
Per 16-bit type: one rocprofv3 pass with --pmc GRBM_GUI_ACTIVE (+ the kernel trace, no other trace domain) over the warm level-0 launch
(tools/proj_bench.py --only pairqo): busy cycles per launch / launch duration = the shader clock the chip sustained under that launch.
    python tools/p3_clock.py [--imgs 64]      (GPU box, repo root) -> one JSON line"""
import argparse, json, os, sqlite3, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--imgs", type=int, default=64)
ap.add_argument("--iters", type=int, default=200)
a = ap.parse_args()
out = {}
for dt in ("fp16", "bf16", "fp16", "bf16"):
    d = "/tmp/p3clk_%s_%d" % (dt, len(out))
    subprocess.run(["rm", "-rf", d])
    r = subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", "GRBM_GUI_ACTIVE", "-d", d, "-o", "k", "--", sys.executable, os.path.join(ROOT, "tools", "proj_bench.py"),
                        "--only", "pairqo", "--iters", str(a.iters), "--rounds", "1", "--imgs", str(a.imgs), "--dtype", dt], cwd="/tmp",
                       env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True)
    db = [os.path.join(dp, f) for dp, _, fs in os.walk(d) for f in fs if f.endswith("_results.db")]
    assert db, r.stderr[-1500:]
    c = sqlite3.connect(db[0])
    tabs = [x[0] for x in c.execute("select name from sqlite_master where type='table'")]
    pick = lambda pre: [t for t in tabs if t.startswith(pre)][0]
    kd, ks, pe = pick("rocpd_kernel_dispatch"), pick("rocpd_info_kernel_symbol"), pick("rocpd_pmc_event")
    kcols = [x[1] for x in c.execute("pragma table_info(%s)" % ks)]
    nc = "display_name" if "display_name" in kcols else "kernel_name"
    rows = list(c.execute("select d.end - d.start, e.value from %s e join %s d on e.event_id = d.event_id join %s s on d.kernel_id = s.id "
                          "where s.%s like '%%proj_p3%%' order by d.start" % (pe, kd, ks, nc)))
    rows = rows[len(rows) // 4:]                      # the launches behind the warm-up quarter
    ns = sum(x[0] for x in rows) / len(rows)
    cyc = sum(x[1] for x in rows) / len(rows)
    out.setdefault(dt, []).append({"launches": len(rows), "avg_us": round(ns / 1e3, 2), "busy_cycles": round(cyc), "GHz": round(cyc / ns, 3)})
print(json.dumps(out))
