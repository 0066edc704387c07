"""Per-dispatch durations [us] of the kernels whose name contains argv[2], from a rocprofv3 .db (views `kernels`)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
names = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
view = "kernels" if "kernels" in names else None
if view is None:
    print("views:", names); sys.exit(0)
cols = [r[1] for r in cur.execute(f"pragma table_info({view})")]
ni, si, ei = cols.index("name"), cols.index("start"), cols.index("end")
rows = [r for r in cur.execute(f"select * from {view} order by start") if sys.argv[2] in r[ni]]
d = [(r[ei] - r[si]) / 1e3 for r in rows]
print(len(d), "dispatches; us:", [round(x) for x in d[: int(sys.argv[3]) if len(sys.argv) > 3 else 80]])
