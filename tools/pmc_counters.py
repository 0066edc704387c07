"""Per-launch averages of arbitrary rocprofv3 --pmc counters for one kernel (text summary for profiles/)."""
import sqlite3, sys
kernel = sys.argv[1]
print(f"kernel filter: {kernel}   (averages per launch)")
for path in sys.argv[2:]:
    db = sqlite3.connect(path); cur = db.cursor()
    for name, calls, val, dur in cur.execute(
            "select counter_name, count(*), avg(value), avg(duration) from counters_collection where kernel_name like ? group by counter_name", (f"%{kernel}%",)):
        print(f"{name}: launches={calls} avg={val:.4g} avg_duration={dur / 1e3:.1f} us")
