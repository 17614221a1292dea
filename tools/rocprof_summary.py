"""Summarises a rocprofv3 (rocpd sqlite) kernel trace: per kernel name x launch geometry,
count / avg / min / max duration.  usage: rocprof_summary.py <results.db> [out.txt]"""
import re
import sqlite3
import sys

db = sys.argv[1]
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
con = sqlite3.connect(db)
cur = con.cursor()
tot = cur.execute("select sum(duration) from kernels").fetchone()[0] or 1
print(f"# rocprofv3 --kernel-trace --stats summary of {db}; total kernel time {tot / 1e6:.3f} ms", file=out)
print(f"{'kernel':58s} {'grid(blocks)':>16s} {'lds':>7s} {'vgpr':>5s} {'calls':>7s} {'total_ms':>9s} {'pct':>6s} {'avg_us':>8s} {'min_us':>8s} {'max_us':>8s}", file=out)
q = """select name, grid_x/workgroup_x, grid_y/workgroup_y, grid_z/workgroup_z, lds_size, vgpr_count, count(*), sum(duration), avg(duration), min(duration), max(duration)
       from kernels group by name, grid_x, grid_y, grid_z, lds_size order by sum(duration) desc"""
for r in cur.execute(q):
    n = re.sub(r"\(.*", "", r[0].replace("(anonymous namespace)::", "")).replace("void ", "")      # (the wide-decode kernels live in an anonymous namespace)
    print(f"{n[:58]:58s} {str((r[1], r[2], r[3])):>16s} {r[4]:7d} {r[5]:5d} {r[6]:7d} {r[7] / 1e6:9.3f} {100.0 * r[7] / tot:6.2f} {r[8] / 1e3:8.2f} {r[9] / 1e3:8.2f} {r[10] / 1e3:8.2f}", file=out)
