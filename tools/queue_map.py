"""Which hardware queue every kernel name ran on, and the k_flow2_lm durations (camera / object launches apart by grid... both 16384: by queue), from a rocprofv3 --kernel-trace .db"""
import glob, os, sqlite3, sys, collections
path = sys.argv[1]
if os.path.isdir(path): path = sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True))[-1]
db = sqlite3.connect(path); cur = db.cursor()
kc = [r[1] for r in cur.execute("pragma table_info(kernels)")]
ni, si, ei, qi = kc.index("name"), kc.index("start"), kc.index("end"), kc.index("queue_id")
rows = list(cur.execute("select * from kernels order by start"))
byq = collections.defaultdict(collections.Counter)
for r in rows: byq[r[qi]][r[ni].split("(")[0][:40]] += 1
for q in sorted(byq): print("queue", q, dict(byq[q].most_common(12)))
lm = collections.defaultdict(list)
for r in rows:
    if "k_flow2_lm" in r[ni]: lm[r[qi]].append((r[ei] - r[si]) / 1e3)
for q, d in lm.items():
    d2 = sorted(d[len(d) // 2:])     # the second half of the run (sync mode runs last)
    print(f"k_flow2_lm on queue {q}: {len(d)} launches, second half median {d2[len(d2)//2]:.0f} us, mean {sum(d2)/len(d2):.0f}, max {d2[-1]:.0f}")
