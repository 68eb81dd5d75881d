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


if __name__ == "__main__":
    print("kernel,calls,total_ns,avg_ns,min_ns,max_ns,grid_x,wg_x")
    for p in sys.argv[1:]:
        rows, _ = stats(p)
        for r in rows:
            print(",".join('"%s"' % r[0][:110] if i == 0 else ("%.0f" % r[i] if isinstance(r[i], float) else str(r[i])) for i in range(len(r))))


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
