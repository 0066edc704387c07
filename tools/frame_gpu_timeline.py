"""GPU-side timeline of ONE frame from a rocprofv3 --kernel-trace --memory-copy-trace .db: every kernel dispatch and memory copy between two consecutive
object-LM launches (the k_flow2_lm dispatches with the larger grid), start offset [us] and duration [us].  argv: db, index of the frame (default 10)."""
import glob, os, sqlite3, sys
path = sys.argv[1]
if os.path.isdir(path): path = sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True))[-1]
db = sqlite3.connect(path); cur = db.cursor()
names = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
def cols(v): return [r[1] for r in cur.execute(f"pragma table_info({v})")]
ev = []
kc = cols("kernels")
ni, si, ei = kc.index("name"), kc.index("start"), kc.index("end")
gi = kc.index("grid_x") if "grid_x" in kc else (kc.index("grid_size_x") if "grid_size_x" in kc else None)
qi = kc.index("queue_id") if "queue_id" in kc else None
for r in cur.execute("select * from kernels"):
    ev.append((r[si], r[ei], r[ni][:64], r[gi] if gi is not None else 0, r[qi] if qi is not None else -1))
mv = [n for n in names if "memory_cop" in n.lower() and not n.startswith("rocpd_")]
if mv:
    mc = cols(mv[0])
    msi, mei = mc.index("start"), mc.index("end")
    mni = mc.index("name") if "name" in mc else None
    mbi = mc.index("size") if "size" in mc else None
    for r in cur.execute(f"select * from {mv[0]}"):
        ev.append((r[msi], r[mei], "COPY " + (str(r[mni]) if mni is not None else "") + (f" {r[mbi]} B" if mbi is not None else ""), 0, -2))
else:
    print("views:", names)
ev.sort()
lm = [i for i, e in enumerate(ev) if "k_flow2_lm" in e[2]]
# object launches: every second LM launch (camera, objects alternate); take pairs by order
k = int(sys.argv[2]) if len(sys.argv) > 2 else 10
a, b = lm[2 * k], lm[2 * k + 2]
t0 = ev[a][0]
for e in ev[a:b + 1]:
    print(f"{(e[0] - t0) / 1e3:9.1f} us +{(e[1] - e[0]) / 1e3:7.1f}  q={e[4]:>3} grid {e[3]:>7}  {e[2]}")
