"""Per-kernel summary of a rocprofv3 --kernel-trace results DB (rocpd sqlite)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'kernel_symbol' in t][0]
q = (f"select s.kernel_name, count(*), avg(d.end-d.start)/1000.0, min(d.end-d.start)/1000.0 from {kd} d "
     f"join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc")
for r in cur.execute(q):
    name = r[0].replace("_ZN4rrtx12_GLOBAL__N_1", "")
    print(f"{r[2]:9.1f} us avg {r[3]:9.1f} min  x{r[1]:4d}  {name[:70]}")
