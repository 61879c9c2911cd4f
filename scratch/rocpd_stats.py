"""Kernel statistics (rocprofv3 --stats equivalent) from a rocprofv3 SQLite result; writes CSV."""
import sqlite3, sys
db = sys.argv[1]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
t = [x for x in tabs if 'kernel_symbol' in x][0]
d = [x for x in tabs if 'kernel_dispatch' in x][0]
rows = list(c.execute(f"select s.kernel_name, count(*), sum(k.end-k.start), avg(k.end-k.start), min(k.end-k.start), max(k.end-k.start) from {d} k join {t} s on k.kernel_id=s.id group by s.kernel_name order by 3 desc"))
tot = sum(r[2] for r in rows)
print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"')
for r in rows:
    name = r[0].split('(')[0][:90]
    print('"%s",%d,%d,%.1f,%.2f,%d,%d' % (name, r[1], r[2], r[3], 100.0 * r[2] / tot, r[4], r[5]))
