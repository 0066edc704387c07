"""Worker of test_ba_gpu.py::test_g2o_operation_order_build_keeps_every_block_at_1e12 (its own process: the product library is chosen at import time
by VDO_HIP_LIB).  Linearises a few synthetic graphs with whatever library the environment names and prints, per block class, the largest deviation from
the oracle's block relative to that class's OWN largest entry - no Cauchy-Schwarz allowance for the right-hand sides - plus the two chi2 deviations."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

BLOCKS = ("Hpp", "bp", "Hll", "bl", "Hpl_eb", "Hll_et", "Hlp1_et", "Hlp2_et", "Hpp_ep")
SHAPES = [(6, 100, 1, 10), (12, 300, 2, 40), (40, 2000, 3, 150), (25, 3000, 0, 0)]


def main():
    from tests import oracle_lib
    from vdo_slam_amd import _capi as K, synth
    from vdo_slam_amd.ba import BatchBA, Context
    o = oracle_lib.load()
    ctx = Context(0)
    worst = {n: 0.0 for n in BLOCKS}
    worst["chi2"] = worst["robust_chi2"] = 0.0
    for shape in SHAPES:
        g = synth.make_ba_graph(*shape, seed=11)
        ba = BatchBA(ctx, g)
        ba.linearize()
        S = ba.system()
        gc, keep = K.graph_to_c(g)
        R = K.BASystem(g)
        assert o.vdo_oracle_ba_linearize(C.byref(gc), C.byref(R.c)) == 0
        for n in BLOCKS:
            a, b = getattr(S, n), getattr(R, n)
            if b.size:
                worst[n] = max(worst[n], float(np.abs(a - b).max() / np.abs(b).max()))
        worst["chi2"] = max(worst["chi2"], abs(S.chi2 - R.chi2) / abs(R.chi2))
        worst["robust_chi2"] = max(worst["robust_chi2"], abs(S.robust_chi2 - R.robust_chi2) / abs(R.robust_chi2))
        ba.close()
    ctx.close()
    print("G2O_ORDER " + json.dumps({"lib": os.path.basename(K.LIB_PATH), "worst": worst}))


if __name__ == "__main__":
    main()
