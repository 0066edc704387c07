"""Host-side Tracking bookkeeping of the product (vdo_dyn_obj_tracking, vdo_tracks_*: no GPU involved)
against the oracle's sequential restatement of the reference (src/Tracking.cc:1366-1612, 2201-2421)."""
import numpy as np
import pytest

from tests import tracking_ref as T
from vdo_slam_amd import tracking as TR


def _dyn_case(seed, f_id, max_id):
    rng = np.random.default_rng(seed)
    sizes = {1: 900, 2: 700, 3: 400, 4: 120, 5: 600, 6: 500, 9: 300}          # label 4 is too small
    sem = np.concatenate([np.full(n, l, np.int32) for l, n in sizes.items()])
    perm = rng.permutation(sem.size)
    sem = sem[perm]
    n = sem.size
    kx = rng.uniform(60, 1180, n).astype(np.float32); ky = rng.uniform(30, 345, n).astype(np.float32)
    b = sem == 5                                                              # label 5 hugs the image border
    kx[b] = rng.uniform(0, 45, b.sum()).astype(np.float32)
    kx[b & (rng.random(n) < 0.3)] = 600.0
    depth = rng.uniform(6, 22, n).astype(np.float32)
    depth[sem == 6] = rng.uniform(24, 40, (sem == 6).sum()).astype(np.float32)  # label 6 is too far
    flow3d = rng.normal(0, 0.6, (n, 3)).astype(np.float32)
    flow3d[sem == 2] *= 0.05                                                  # label 2 does not move: static
    obj_label = np.where(rng.random(n) < 0.08, -1, 1).astype(np.int32)        # some outliers from the previous stage
    last_sem = sem.copy()
    flip = rng.random(n) < 0.2
    last_sem[flip] = rng.integers(0, 10, flip.sum())
    last_sem[sem == 9] = 0                                                    # label 9 was background last time: vote says 0
    prm = TR.DynObjParamsC(1242, 375, 25, 50, 0.12, 0.3, 25.0, f_id)
    last_tab = (np.array([1, 3, 0, 2], np.int32), np.array([4, 7, 9, 5], np.int32), np.array([1, 0, 1, 1], np.uint8))
    return prm, sem, obj_label, kx, ky, depth, flow3d, last_sem, last_tab, max_id


@pytest.mark.parametrize("seed,f_id,max_id", [(1, 5, 10), (2, 1, 77), (3, 9, 4)])
def test_dyn_obj_tracking_matches_oracle(oracle, seed, f_id, max_id):
    prm, sem, ol, kx, ky, depth, fl, last_sem, tab, mid = _dyn_case(seed, f_id, max_id)
    got = TR.dyn_obj_tracking(prm, sem, ol, kx, ky, depth, fl, last_sem, *tab, mid)
    exp = T.dyn_obj_tracking(oracle, prm, sem, ol, kx, ky, depth, fl, last_sem, *tab, mid)
    assert np.array_equal(got["obj_label"], exp["obj_label"])
    assert len(got["objects"]) == len(exp["objects"]) and got["max_id"] == exp["max_id"]
    for a, b in zip(got["objects"], exp["objects"]):
        assert np.array_equal(a, b)
    assert np.array_equal(got["sem"], exp["sem"]) and np.array_equal(got["mod"], exp["mod"])
    # the scenario really exercises every branch
    lab = got["obj_label"]
    assert set(got["sem"].tolist()) == {1, 3, 9}
    assert np.all(lab[(sem == 2) & (ol != -1)] == 0)                           # static object
    assert np.all(lab[sem == 4] == -1) and np.all(lab[sem == 5] == -1) and np.all(lab[sem == 6] == -1)
    if f_id != 1:
        assert got["mod"][0] == 4                                             # label 1 -> tracked object with motion label 4
        assert got["mod"][1] == mid                                           # label 3's last object was not tracked: new id
        assert got["mod"][2] == 9                                             # vote says "background (0)" -> last object with sem 0
    else:
        assert got["mod"].tolist() == [1, 9, 2] or got["mod"][0] == 1


def _asso_sequence(seed, n_frames, with_label):
    rng = np.random.default_rng(seed)
    assos, labels = [], []
    n_prev = 0
    for i in range(n_frames):
        n = int(rng.integers(5, 400))
        if i == 0:
            a = np.where(rng.random(n) < 0.8, rng.integers(0, 500, n), -1)    # frame 0 matches into the (virtual) first frame
        else:
            a = np.full(n, -1)
            k = min(n, n_prev)
            src = rng.permutation(n_prev)[:k]                                 # injective: a feature continues at most one track
            sel = rng.permutation(n)[:k]
            keep = rng.random(k) < 0.7
            a[sel[keep]] = src[keep]
        assos.append(a.astype(np.int32))
        labels.append(rng.integers(1, 6, n).astype(np.int32))
        n_prev = n
    return assos, (labels if with_label else None)


@pytest.mark.parametrize("seed,n_frames,with_label", [(1, 12, False), (2, 30, True), (3, 1, True), (4, 2, False)])
def test_incremental_tracks_equal_rebuild_from_scratch(oracle, seed, n_frames, with_label):
    assos, labels = _asso_sequence(seed, n_frames, with_label)
    tb = TR.TrackBuilder(with_label)
    for i, a in enumerate(assos):
        tb.add_frame(a, labels[i] if with_label else None)
        if i in (0, n_frames // 2, n_frames - 1):
            # the reference rebuilds everything from frame 0 on every frame: same tracklets, same order
            off, pf, pt, oid = tb.get()
            eoff, epf, ept, eoid = T.build_tracks(oracle, assos[:i + 1], labels[:i + 1] if with_label else None)
            assert np.array_equal(off, eoff) and np.array_equal(pf, epf) and np.array_equal(pt, ept)
            if with_label:
                assert np.array_equal(oid, eoid)
    off, pf, pt, _ = tb.get()
    assert off.size - 1 > 0 and np.all(np.diff(off) >= 2)
    # every tracklet is a run of consecutive frames
    for t in range(min(off.size - 1, 200)):
        fr = pf[off[t]:off[t + 1]]
        assert np.array_equal(fr, np.arange(fr[0], fr[0] + fr.size))


def test_tracks_reject_out_of_range_association():
    from vdo_slam_amd import _capi as K
    tb = TR.TrackBuilder(False)
    tb.add_frame(np.array([0, -1, 2], np.int32))
    with pytest.raises(K.VdoError):
        tb.add_frame(np.array([7], np.int32))
