"""Where a window-sized batch LM spends an iteration (20 poses, ~2.3 k points: launch-bound): run under
   rocprofv3 --kernel-trace -d DIR -- python tools/window_lm_timeline.py run
then  python tools/window_lm_timeline.py report DIR  prints, for the LAST optimize call, wall span, summed kernel time and the per-kernel list of one iteration."""
import glob
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    import time
    from vdo_slam_amd import synth
    from vdo_slam_amd.ba import BatchBA, Context
    shape = tuple(int(x) for x in os.environ.get("VDO_PROBE_SHAPE", "20,2200,1,20").split(","))
    g = synth.make_ba_graph(*shape, seed=3)
    print("P", g.n_pose, "L", g.n_point, "Eb", g.n_eb, "Et", g.n_et, "Ep", g.n_ep)
    ctx = Context(0)
    for rep in range(3):
        ba = BatchBA(ctx, g)
        t = time.perf_counter()
        st = ba.optimize(max_iterations=int(os.environ.get("VDO_PROBE_ITS", "12")), gain_threshold=-1.0, verbose=0)
        dt = time.perf_counter() - t
        print("LM its", st.iterations, "trials", st.total_trials, "ms", dt * 1e3, "ms/trial", dt * 1e3 / st.total_trials)
        ba.close()
    ctx.close()


def report(d):
    db = sqlite3.connect(sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True))[-1]); cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    ni, si, ei = cols.index("name"), cols.index("start"), cols.index("end")
    rows = list(cur.execute("select * from kernels order by start"))
    # an optimize call starts with k_max_diag (computeLambdaInit); argv[3] = which one (default: the last), up to the next frame kernel
    starts = [i for i, r in enumerate(rows) if "k_max_diag" in r[ni]]
    which = int(sys.argv[3]) if len(sys.argv) > 3 else -1
    print(len(starts), "optimize calls in the trace")
    i0 = starts[which]
    i1 = i0
    while i1 < len(rows) and not any(k in rows[i1][ni] for k in ("k_flow2_lm", "k_pyramid", "k_ingest")) and not (i1 > i0 and "k_max_diag" in rows[i1][ni]): i1 += 1
    rows = rows[i0:i1]
    span = (rows[-1][ei] - rows[0][si]) / 1e3
    busy = sum(r[ei] - r[si] for r in rows) / 1e3
    nfac = sum(1 for r in rows if "k_factor_chains" in r[ni])
    print(f"last optimize: {len(rows)} dispatches, {nfac} trials, span {span:.0f} us, kernels {busy:.0f} us ({busy / span:.2f}); per trial: {len(rows) / max(nfac, 1):.1f} dispatches, {span / max(nfac, 1):.0f} us")
    # one trial in the middle
    f = [i for i, r in enumerate(rows) if "k_factor_chains" in r[ni]]
    if len(f) > 3:
        a, b = f[2], f[3]
        t0 = rows[a][si]
        for r in rows[a:b]:
            print(f"{(r[si] - t0) / 1e3:8.1f} us +{(r[ei] - r[si]) / 1e3:6.1f}  {r[ni][:70]}")


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else report(sys.argv[2])
