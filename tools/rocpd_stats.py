"""Per-kernel summary of a rocprofv3 rocpd database (the default output of `rocprofv3 --kernel-trace`):
python tools/rocpd_stats.py <results.db> [top]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows = list(db.execute("select name, count(*), avg(end-start)/1e6, min(end-start)/1e6, max(end-start)/1e6, sum(end-start)/1e6 "
                       "from kernels group by name order by 6 desc"))
total = sum(r[5] for r in rows)
print(f"{'kernel':<80} {'calls':>6} {'avg ms':>10} {'min ms':>10} {'max ms':>10} {'total ms':>10} {'%':>6}")
for r in rows[:top]:
    print(f"{r[0][:80]:<80} {r[1]:>6} {r[2]:>10.3f} {r[3]:>10.3f} {r[4]:>10.3f} {r[5]:>10.2f} {100 * r[5] / total:>6.1f}")
