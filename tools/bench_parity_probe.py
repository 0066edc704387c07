"""GPU probe: the bench sequence (KITTI-0000 length by default) through the reference itself (oracle/_ref/libref_full.so, child process) and through the product's
System::TrackRGBD; prints the parity record and keeps both per-frame records.  usage: python tools/bench_parity_probe.py [steps] [warmup] [out_dir]"""
import json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vdo_slam_amd import synth_seq as SQ
from tests import bench_parity as BP

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 148
warmup = int(sys.argv[2]) if len(sys.argv) > 2 else 5
out_dir = sys.argv[3] if len(sys.argv) > 3 else "gpurun_out/parity"
os.makedirs(out_dir, exist_ok=True)
spec = SQ.bench_spec(warmup, steps)
labels = BP.labels_of(spec)
with tempfile.TemporaryDirectory(dir="/tmp") as td:
    t0 = time.time()
    frames = SQ.render_bench_sequence(spec, os.path.join(td, "frames"))
    print(f"rendered {len(frames)} frames in {time.time() - t0:.1f} s", flush=True)
    cfg = BP.write_bench_settings(os.path.join(td, "kitti.yaml"))
    ref_npz = os.path.join(td, f"ref_{steps}_{warmup}.npz")
    t0 = time.time()
    proc = BP.start_reference(cfg, os.path.join(td, "frames"), len(frames), ref_npz, labels)
    got = BP.product_sequence(cfg, frames, labels)
    print(f"product sequence done in {time.time() - t0:.1f} s", flush=True)
    from tests.ref_track import finish_sequence
    ref = finish_sequence(proc, ref_npz)
    ref = {q: ref[q] for q in ref.files}
    print(f"reference done after {time.time() - t0:.1f} s (its own clock: {float(ref['seconds']):.1f} s)", flush=True)
par = BP.compare(ref, got)
print(json.dumps(par, indent=1))
print("assert_parity:", BP.assert_parity(par))
json.dump(par, open(os.path.join(out_dir, f"parity_{steps}_{warmup}.json"), "w"), indent=1)
