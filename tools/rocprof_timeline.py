"""Prints a window of consecutive kernel dispatches (start offset, duration, queue/stream) from a rocprofv3
rocpd database: shows whether launches overlap.  usage: rocprof_timeline.py <results.db> [first] [count]"""
import re, sqlite3, sys
db = sys.argv[1]
first = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
count = int(sys.argv[3]) if len(sys.argv) > 3 else 40
cur = sqlite3.connect(db).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = list(cur.execute(f"select name, start, end{', ' + qcol if qcol else ''} from kernels order by start limit {count} offset {first}"))
t0 = rows[0][1]
prev_end = t0
for r in rows:
    n = re.sub(r"\(.*", "", r[0]).replace("void ", "")[:44]
    print(f"{n:44s} q={r[3] if qcol else '-'} start {(r[1]-t0)/1e3:9.2f} us  dur {(r[2]-r[1])/1e3:7.2f} us  gap-to-prev-end {(r[1]-prev_end)/1e3:7.2f}")
    prev_end = r[2]
