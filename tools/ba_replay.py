"""Replay a `.g2o` dump of the reference (or of this repo) through the batch optimiser and write the
optimised graph next to it — the third-party parity route of SURVEY.md §8f-1: anyone with a real build of
the reference can diff its `after_opt` dump against ours.

    python tools/ba_replay.py graph.g2o [--out after_opt.g2o] [--iterations 300] [--gain 1e-4] [--oracle]

--oracle runs the CPU oracle instead of the GPU (no GPU needed)."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vdo_slam_amd import _capi as K  # noqa: E402
from vdo_slam_amd import g2o_io  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("graph")
    ap.add_argument("--out")
    ap.add_argument("--iterations", type=int, default=300)
    ap.add_argument("--gain", type=float, default=1e-4)
    ap.add_argument("--huber", type=float, default=1e-4)
    ap.add_argument("--oracle", action="store_true")
    a = ap.parse_args()
    g = g2o_io.read_g2o(a.graph, a.huber, a.huber, a.huber)
    print(f"{a.graph}: {g.n_pose} SE3 vertices, {g.n_point} points, {g.n_eb} EDGE_SE3_TRACKXYZ, {g.n_et} EDGE_SE3_MOTION, {g.n_ep} EDGE_SE3:QUAT, {g.n_prior} priors")
    if a.oracle:
        from tests import oracle_lib
        o = oracle_lib.load()
        gc, keep = K.graph_to_c(g)
        opt = K.LMOptionsC(a.iterations, a.gain, 1, 0, 0.0, 0)
        st = K.LMStatsC()
        pose = np.zeros_like(g.pose); point = np.zeros_like(g.point)
        o.vdo_oracle_ba_optimize(C.byref(gc), C.byref(opt), K._dp(pose), K._dp(point), C.byref(st))
    else:
        from vdo_slam_amd.ba import BatchBA, Context
        ba = BatchBA(Context(0), g)
        st = ba.optimize(max_iterations=a.iterations, gain_threshold=a.gain, verbose=1)
        pose, point = ba.estimates()
    print(f"iterations {st.iterations}, trials {st.total_trials}, chi2 {st.initial_chi2:.9g} -> {st.final_chi2:.9g}, stop reason {st.stop_reason}")
    out = a.out or os.path.splitext(a.graph)[0] + "_after_opt.g2o"
    g2o_io.write_g2o(out, g, pose, point)
    print("wrote", out)


if __name__ == "__main__":
    main()
