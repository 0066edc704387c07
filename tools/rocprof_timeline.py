"""Timeline of kernel dispatches from a rocprofv3 .db: prints name, start offset [us] and duration [us] for dispatches in
[t0, t1) us after the argv[2]-th k_flow2_lm dispatch (debug aid for the per-frame overlap)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
ni, si, ei = cols.index("name"), cols.index("start"), cols.index("end")
qi = cols.index("queue_id") if "queue_id" in cols else (cols.index("stream_id") if "stream_id" in cols else None)
rows = list(cur.execute("select * from kernels order by start"))
lm = [r for r in rows if "flow2" in r[ni]]
ref = lm[int(sys.argv[2])][si]
span = float(sys.argv[3]) if len(sys.argv) > 3 else 4000.0
for r in rows:
    t = (r[si] - ref) / 1e3
    if -50 <= t < span:
        print(f"{t:9.1f} us  +{(r[ei] - r[si]) / 1e3:8.1f} us  q={r[qi] if qi is not None else '?'}  {r[ni][:60]}")
