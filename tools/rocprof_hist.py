"""Duration histogram of one kernel in a rocprofv3 (rocpd sqlite) kernel trace, and its mean by occurrence index modulo
`period` (8 = decode steps per captured graph: shows whether the outliers sit at graph boundaries).
usage: rocprof_hist.py <results.db> <kernel name substring> [period] [skip first n]"""
import sqlite3, sys
db, sub = sys.argv[1], sys.argv[2]
period = int(sys.argv[3]) if len(sys.argv) > 3 else 8
skip = int(sys.argv[4]) if len(sys.argv) > 4 else 0
cur = sqlite3.connect(db).cursor()
rows = [(s, e - s) for s, e in cur.execute("select start, end from kernels where name like ? order by start", (f"%{sub}%",))][skip:]
d = [x[1] / 1e3 for x in rows]
print(f"# {sub}: {len(d)} launches, mean {sum(d) / max(1, len(d)):.2f} us")
edges = [0, 4, 8, 12, 14, 16, 18, 20, 24, 28, 32, 40, 50, 60, 80, 1e9]
for lo, hi in zip(edges[:-1], edges[1:]):
    n = sum(1 for v in d if lo <= v < hi)
    if n:
        print(f"  {lo:5.0f} .. {hi if hi < 1e8 else float('inf'):5.0f} us : {n:6d}  {'#' * min(60, (60 * n) // len(d) + 1)}")
for r in range(period):
    v = d[r::period]
    print(f"  index % {period} == {r}: n {len(v):5d}  mean {sum(v) / max(1, len(v)):7.2f} us  max {max(v) if v else 0:7.2f}")
# gap between this kernel's start and the previous kernel's end (any kernel)
allk = list(cur.execute("select start, end, name from kernels order by start"))
gaps = []
for i in range(1, len(allk)):
    if sub in allk[i][2]:
        gaps.append((allk[i][0] - allk[i - 1][1]) / 1e3)
gaps = gaps[skip:]
if gaps:
    print(f"  gap to the previous kernel's end: mean {sum(gaps) / len(gaps):.2f} us, max {max(gaps):.2f} us")
