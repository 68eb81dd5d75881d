"""Summarise a rocprofv3 rocpd SQLite database (kernel-trace) into a per-kernel stats table.

usage: python tools/rocpd_stats.py results.db [more.db ...]   -> CSV on stdout
(`rocprofv3 --output-format csv --stats` gives the same numbers; this works on the default db too.)
"""
import sqlite3
import sys


def stats(path):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in c.execute("pragma table_info(%s)" % kd)]
    kcols = [r[1] for r in c.execute("pragma table_info(%s)" % ks)]
    name_col = "display_name" if "display_name" in kcols else "kernel_name"
    q = ("select s.%s, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start), "
         "d.grid_size_x, d.workgroup_size_x from %s d join %s s on d.kernel_id = s.id group by s.%s, d.grid_size_x "
         "order by 3 desc" % (name_col, kd, ks, name_col))
    return list(c.execute(q)), cols


def pmc_stats(path):
    """Per (kernel, grid) average of every collected counter: rows (kernel, grid_x, counter, calls, avg)."""
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    pick = lambda pre: [t for t in tabs if t.startswith(pre)][0]
    kd, ks, pe, ip = pick("rocpd_kernel_dispatch"), pick("rocpd_info_kernel_symbol"), pick("rocpd_pmc_event"), pick("rocpd_info_pmc")
    kcols = [r[1] for r in c.execute("pragma table_info(%s)" % ks)]
    name_col = "display_name" if "display_name" in kcols else "kernel_name"
    q = ("select s.%s, d.grid_size_x, i.name, count(*), avg(e.value) from %s e join %s d on e.event_id = d.event_id "
         "join %s s on d.kernel_id = s.id join %s i on e.pmc_id = i.id group by s.%s, d.grid_size_x, i.name order by 5 desc"
         % (name_col, pe, kd, ks, ip, name_col))
    return list(c.execute(q))


if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[1] == "--pmc":
    # usage: rocpd_stats.py --pmc results.db  -> kernel,grid_x,counter,calls,avg
    for r in pmc_stats(sys.argv[2]):
        print("%s,%d,%s,%d,%.1f" % (r[0][:100], r[1], r[2], r[3], r[4]))
    sys.exit(0)

if __name__ == "__main__" and not (len(sys.argv) > 1 and sys.argv[1] == "--breakdown"):
    print("kernel,calls,total_ns,avg_ns,min_ns,max_ns,grid_x,wg_x")
    for p in sys.argv[1:]:
        rows, _ = stats(p)
        for r in rows:
            print(",".join('"%s"' % r[0][:110] if i == 0 else ("%.0f" % r[i] if isinstance(r[i], float) else str(r[i])) for i in range(len(r))))


def category(n):
    if "naive_conv" in n:
        return "miopen-find (naive conv, warm-up only)"
    if "xattn" in n or "pack_kv" in n or "pack_wq" in n or "dcoef_reduce" in n:
        return "sta xattn (ours)"
    if "selfattn_fwd" in n:
        return "sta self-attention (ours)"
    if "gemm_rows_kernel" in n or "pack_gemm_w" in n:
        return "sta row GEMM: proj_in / proj_out + residual, skip 1x1 over the concatenation (ours)"
    if "conv3x3_nhwc_kernel" in n or "pack_conv_w" in n:
        return "sta 3x3 convolution (ours)"
    if "to_out_ln_ofrag" in n or "ff_geglu_qfrag" in n or "ff_out_res_hfrag" in n or "add_layernorm_qfrag" in n or "pack_w" in n:
        return "sta level-0 chain: to_out+LN, GEGLU projection, ff output, fragment LayerNorm (ours)"
    if "stats_finalize_kernel" in n or "gn_silu_kernel" in n or "geglu_kernel" in n or "add_layernorm_kernel" in n or "add_bias_nchw_kernel" in n or "gn_nhwc_" in n or "add_bias_rows_kernel" in n:
        return "sta trunk glue: GroupNorm/GEGLU/LayerNorm/residual (ours)"
    if "attn_fwd" in n or "attention" in n.lower():
        return "SDPA self-attention (d = 160 levels)"
    if "igemm" in n or "conv" in n.lower() or "GridwiseGemm" in n and "grouped_conv" in n:
        return "conv (MIOpen/CK)"
    if "Cijk" in n or "gemm" in n.lower():
        return "GEMM (hipBLASLt/rocBLAS)"
    if "RowwiseMoments" in n or "GroupNorm" in n or "group_norm" in n or "ComputeFused" in n:
        return "GroupNorm"
    if "layer_norm" in n.lower():
        return "LayerNorm"
    if "transpose" in n.lower():
        return "transpose (layout)"
    if "SubTensor" in n or "OpTensor" in n:
        return "MIOpen tensor ops (bias add ...)"
    if "elementwise" in n.lower() or "copy" in n.lower():
        return "elementwise / copy (aten)"
    return "other"


def breakdown(path, top=12):
    import collections
    rows, _ = stats(path)
    t, c = collections.Counter(), collections.Counter()
    for r in rows:
        t[category(r[0])] += r[2]
        c[category(r[0])] += r[1]
    tot = sum(v for k, v in t.items() if "warm-up" not in k)
    print("category,total_ms,calls,share_of_steady_state")
    for k, v in t.most_common():
        print("%s,%.1f,%d,%.1f%%" % (k, v / 1e6, c[k], 100.0 * v / tot))
    print("--- top kernels (steady state)")
    for r in [r for r in rows if "naive_conv" not in r[0]][:top]:
        print("%.1f ms  %6d calls  avg %.1f us  grid %s  %s" % (r[2] / 1e6, r[1], r[3] / 1e3, r[6], r[0][:120]))


if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[1] == "--breakdown":
    breakdown(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 12)
    sys.exit(0)
