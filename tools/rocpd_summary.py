"""Per-kernel totals from a rocprofv3 rocpd database (the default output when no --output-format is given)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); c = db.cursor()
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'info_kernel_symbol' in t][0]
q = f"select s.kernel_name, d.grid_size_x/d.workgroup_size_x, d.grid_size_y, count(*), sum(d.end-d.start)/1e6, avg(d.end-d.start)/1e3 from {kd} d join {ks} s on d.kernel_id=s.id group by 1,2,3 order by 5 desc limit {int(sys.argv[2]) if len(sys.argv) > 2 else 20}"
for r in c.execute(q):
    print(f"{r[0][:72]:72s} grid=({r[1]},{r[2]}) n={r[3]:5d} total={r[4]:9.3f} ms avg={r[5]:9.2f} us")
