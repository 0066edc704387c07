"""Summarise a rocprofv3 results .db (or kernel_stats csv) into a small text table for profiles/."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(top_kernels)")]
rows = cur.execute("select * from top_kernels").fetchall()
print(" | ".join(cols))
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    print(" | ".join(str(x) if not isinstance(x, float) else f"{x:.3f}" for x in r))
