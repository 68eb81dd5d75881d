"""Why does the fp16 build of a kernel run a few % behind the bf16 one on an instruction stream that is the same line for line?
Per kernel and 16-bit type: one rocprofv3 pass with --pmc GRBM_GUI_ACTIVE (+ the kernel trace, no other trace domain) over a warm series
of launches: busy cycles per launch / launch duration = the shader clock the chip sustained under that launch.
    python tools/p3_clock.py [--kernel p3|selfattn] [--imgs 64]      (GPU box, repo root) -> one JSON line
p3        the level-0 cross-attention launch (head-pair kernel, tools/proj_bench.py --only pairqo)
selfattn  the level-0 self-attention forward (tools/selfattn_l0_time.py, 64 x 8 heads, N = 4096, logits that flag nothing)"""
import argparse, json, os, sqlite3, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--kernel", choices=["p3", "selfattn"], default="p3")
ap.add_argument("--imgs", type=int, default=64)
ap.add_argument("--iters", type=int, default=200)
a = ap.parse_args()
out = {}
for i, dt in enumerate(("fp16", "bf16", "fp16", "bf16")):
    d = "/tmp/kclk_%s_%d" % (dt, i)
    subprocess.run(["rm", "-rf", d])
    env = dict(os.environ, TMPDIR="/tmp")
    if a.kernel == "p3":
        cmd, like = [os.path.join(ROOT, "tools", "proj_bench.py"), "--only", "pairqo", "--iters", str(a.iters), "--rounds", "1", "--imgs", str(a.imgs), "--dtype", dt], "proj_p3"
    else:
        cmd, like = [os.path.join(ROOT, "tools", "selfattn_l0_time.py")], "selfattn_fwd_pipe_kernel%Lb1E"      # the optimistic loop (its repair launch returns at once)
        env.update(STA_SA_DTYPE=dt, STA_SA_QSCALE="0.25")
    r = subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", "GRBM_GUI_ACTIVE", "-d", d, "-o", "k", "--", sys.executable] + cmd, cwd="/tmp", env=env,
                       capture_output=True, text=True)
    db = [os.path.join(dp, f) for dp, _, fs in os.walk(d) for f in fs if f.endswith("_results.db")]
    assert db, r.stderr[-1500:]
    c = sqlite3.connect(db[0])
    tabs = [x[0] for x in c.execute("select name from sqlite_master where type='table'")]
    pick = lambda pre: [t for t in tabs if t.startswith(pre)][0]
    kd, ks, pe = pick("rocpd_kernel_dispatch"), pick("rocpd_info_kernel_symbol"), pick("rocpd_pmc_event")
    kcols = [x[1] for x in c.execute("pragma table_info(%s)" % ks)]
    nc = "display_name" if "display_name" in kcols else "kernel_name"
    rows = list(c.execute("select d.end - d.start, e.value from %s e join %s d on e.event_id = d.event_id join %s s on d.kernel_id = s.id "
                          "where s.%s like '%%%s%%' order by d.start" % (pe, kd, ks, nc, like)))
    rows = rows[len(rows) // 4:]                      # the launches behind the warm-up quarter
    ns = sum(x[0] for x in rows) / len(rows)
    cyc = sum(x[1] for x in rows) / len(rows)
    out.setdefault(dt, []).append({"launches": len(rows), "avg_us": round(ns / 1e3, 2), "busy_cycles": round(cyc), "GHz": round(cyc / ns, 3)})
print(json.dumps({"kernel": a.kernel, **out}))
