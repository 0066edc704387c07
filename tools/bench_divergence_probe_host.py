"""GPU probe: the product's System::TrackRGBD (host buffers) against the oracle-composed Track() on the bench sequence: first differences in detail.
usage: python tools/bench_divergence_probe_host.py [steps] [warmup]"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vdo_slam_amd import synth_seq as SQ
from vdo_slam_amd.system import System
from tests import oracle_lib, bench_parity as BP
from tests.pipeline_ref import OraclePipeline
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
warmup = int(sys.argv[2]) if len(sys.argv) > 2 else 5
spec = SQ.bench_spec(warmup, steps)
labels = BP.labels_of(spec)
td = tempfile.mkdtemp(dir="/tmp")
frames = SQ.render_bench_sequence(spec, os.path.join(td, "f"))
cfg = BP.write_bench_settings(os.path.join(td, "kitti.yaml"))
sysm = System(cfg)
ref = OraclePipeline(oracle_lib.load(), build_lm=True)
ndiff = 0
for k, fr in enumerate(frames):
    depth = fr["depth_raw"].copy(); mask = fr["mask"].copy()
    rows = np.array([[k, lab, 0, 0, 0, 0, 0, 0, 0, 0] for lab in labels], np.float32)
    T = sysm.track_rgbd(fr["gray"], depth, fr["flow"], mask, rows)
    exp = ref.step(fr)
    L = ref.last
    bad = []
    if not np.array_equal(T, ref.Tl): bad.append(f"pose {np.abs(T - ref.Tl).max():.2e}")
    if not np.array_equal(mask, L["mask"]):
        dm = mask != L["mask"]
        pairs, cnt = np.unique(np.stack([mask[dm], L["mask"][dm]], 1), axis=0, return_counts=True)
        bad.append(f"mask: {int(dm.sum())} px differ, (got, exp) x count: {[(int(a), int(b), int(c)) for (a, b), c in zip(pairs, cnt)]}; input labels {np.unique(fr['mask']).tolist()} recovered exp {exp['n_recovered_masks']}")
    n, s = sysm.frame_state(1, 12)
    ob = s.reshape(12, n)
    if n != L["ob"]["key_x"].size:
        bad.append(f"object set size {n} vs {L['ob']['key_x'].size}; sem labels got {np.unique(ob[10].astype(int), return_counts=True)} exp {np.unique(L['ob']['label'], return_counts=True)}")
    elif not np.array_equal(ob[0], L["ob"]["key_x"]):
        bad.append("object set keys")
    n3, s3 = sysm.frame_state(3, 8)
    if n3 != exp["n_object_samples"]:
        sm = s3.reshape(8, n3)
        bad.append(f"samples {n3} vs {exp['n_object_samples']}: got labels {np.unique(sm[7].astype(int), return_counts=True)}")
    n2, s2 = sysm.frame_state(2, 19)
    po = s2.reshape(n2, 19)
    print(f"frame {k}: objects got sem {po[:, 0].astype(int).tolist()} mod {po[:, 1].astype(int).tolist()} stat {po[:, 2].astype(int).tolist()} | exp sem {list(L['sem_pos'])} mod {list(L['mod'])} stat {list(np.asarray(L['stat']).astype(int))} {'<-- DIFF ' + ' | '.join(bad) if bad else ''}", flush=True)
    if bad:
        ndiff += 1
        if ndiff >= 3:
            break
sysm.close()
