"""KATs that pin the front-end oracle (the reference has no tests and OpenCV 3.4 is not
available: parity with OpenCV itself stays UNPINNED, see oracle/frontend_oracle.cpp)."""
import numpy as np
import pytest

from tests import frontend_ref as R
from vdo_slam_amd import synth_frames as SF

RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


@pytest.fixture(scope="module")
def frame():
    return SF.make_frame(seed=3)


def test_level_sizes_and_feature_budget(oracle):
    ws, hs, nf = R.level_sizes(oracle, 1242, 375)
    assert list(ws[:3]) == [1242, 1035, 862] and hs[0] == 375
    assert hs[1] == 312            # 375/1.2 = 312.5 -> cvRound = round-half-even
    assert list(nf) == [543, 452, 377, 314, 262, 218, 182, 152] and nf.sum() == 2500


def test_pyramid_resize_close_to_float_bilinear_and_border_reflects(oracle, frame):
    lv = R.pyramid(oracle, frame["gray"])
    g0 = lv[0][19:-19, 19:-19]
    assert np.array_equal(g0, frame["gray"])
    # border = reflect-101 of the interior
    assert np.array_equal(lv[0][19:-19, 18], frame["gray"][:, 1]) and np.array_equal(lv[0][0, 19:-19], frame["gray"][19, :])
    # level 1 vs float bilinear (half-pixel centres): fixed-point result within 1 LSB
    src = frame["gray"].astype(np.float64)
    l1 = lv[1][19:-19, 19:-19]
    h1, w1 = l1.shape
    sx = (np.arange(w1) + 0.5) * (1242 / w1) - 0.5; sy = (np.arange(h1) + 0.5) * (375 / h1) - 0.5
    x0 = np.clip(np.floor(sx).astype(int), 0, 1240); y0 = np.clip(np.floor(sy).astype(int), 0, 373)
    fx = np.clip(sx - x0, 0, 1)[None, :]; fy = np.clip(sy - y0, 0, 1)[:, None]
    ref = (src[y0][:, x0] * (1 - fx) + src[y0][:, x0 + 1] * fx) * (1 - fy) + (src[y0 + 1][:, x0] * (1 - fx) + src[y0 + 1][:, x0 + 1] * fx) * fy
    assert np.abs(l1.astype(np.float64) - ref).max() <= 1.01


def _brute_corner(img, x, y, t):
    v = int(img[y, x]); r = [int(img[y + dy, x + dx]) for dx, dy in RING]
    for s in range(16):
        arc = [r[(s + k) % 16] for k in range(9)]
        if all(q > v + t for q in arc) or all(q < v - t for q in arc):
            return True
    return False


def test_fast_candidates_are_true_fast9_corners_and_nms_maxima(oracle, frame):
    gray = frame["gray"]
    x, y, r = R.fast_level(oracle, gray, 0)
    assert x.size > 500
    xi = x.astype(int) + 16; yi = y.astype(int) + 16
    # every candidate passes the segment test at the fallback threshold and its score is consistent
    for k in range(0, x.size, 7):
        assert _brute_corner(gray, xi[k], yi[k], 7)
        assert _brute_corner(gray, xi[k], yi[k], int(r[k])) and not _brute_corner(gray, xi[k], yi[k], int(r[k]) + 1)
    # inside the 16-px border
    assert xi.min() >= 19 and xi.max() <= 1242 - 20 and yi.min() >= 19 and yi.max() <= 375 - 20


def test_extract_counts_order_and_angles(oracle, frame):
    kp = R.extract(oracle, frame["gray"])
    n = kp["x"].size
    assert 1500 < n <= 2500 + 8 * 3
    assert np.all(np.diff(kp["octave"]) >= 0)                       # levels concatenated 0..7
    assert np.all((kp["angle"] >= 0) & (kp["angle"] <= 360))
    lv0 = kp["octave"] == 0
    assert np.all(kp["x"][lv0] == np.round(kp["x"][lv0]))           # level-0 coordinates are integers
    assert kp["size"][0] == 31 and kp["size"][-1] == int(31 * np.float32(1.2) ** 7)


def test_blur_matches_float_gaussian_within_fixed_point_gain(oracle, frame):
    img = frame["gray"][:120, :200].copy()
    out = R.blur7(oracle, img).astype(np.float64)
    k = np.exp(-0.5 * (np.arange(7) - 3) ** 2 / 4.0); k /= k.sum()
    pad = np.pad(img.astype(np.float64), 3, mode="reflect")
    tmp = sum(k[i] * pad[:, i:i + img.shape[1]] for i in range(7))
    ref = sum(k[i] * tmp[i:i + img.shape[0], :] for i in range(7))
    # 8-bit fixed-point kernel (55,49,34,18) sums to 257/256 per pass -> gain (257/256)^2
    # (each 8-bit weight is off by up to 0.5/256, so a few grey levels of slack on top of the gain)
    assert np.abs(out - ref * (257 / 256) ** 2).max() <= 3.0
    assert np.abs(out - ref * (257 / 256) ** 2).mean() <= 0.6


def test_frame_filters_against_numpy(oracle, frame):
    from vdo_slam_amd.synth_frames import BF, DEPTH_MAP_FACTOR, TH_DEPTH_BG, TH_DEPTH_OBJ
    depth = frame["depth_raw"].copy()
    oracle.vdo_oracle_depth_preprocess(R._fp(depth), depth.size, BF, DEPTH_MAP_FACTOR)
    raw = frame["depth_raw"]
    with np.errstate(divide="ignore"):
        expect = np.where(raw < 0, 0, np.float32(BF) / (raw / np.float32(DEPTH_MAP_FACTOR))).astype(np.float32)
    assert np.array_equal(depth, expect)
    kp = R.extract(oracle, frame["gray"])
    sf = R.static_filter(oracle, kp["x"], kp["y"], kp["octave"], frame["mask"], depth, frame["flow"], TH_DEPTH_BG)
    xi = kp["x"].astype(int); yi = kp["y"].astype(int)
    d = depth[yi, xi]; fl = frame["flow"][yi, xi]
    keep = (frame["mask"][yi, xi] == 0) & ~((d > TH_DEPTH_BG) | (d <= 0)) & (fl[:, 0] != 0) & (fl[:, 1] != 0) & \
           (kp["x"] + fl[:, 0] < 1242) & (kp["y"] + fl[:, 1] < 375)
    assert np.array_equal(sf["keep_idx"], np.nonzero(keep)[0])
    ob = R.object_sample(oracle, frame["mask"], depth, frame["flow"], TH_DEPTH_OBJ)
    assert ob["label"].size > 100 and np.all(ob["label"] > 0)
    assert np.all(np.diff(ob["key_y"] * 4096 + ob["key_x"]) > 0)    # raster order


def test_sincos_exact_is_the_correctly_rounded_float(oracle):
    """The +,-,*-only sin/cos the descriptor stage uses on both sides: equal to the double-precision value rounded to float
    (what a correctly rounded cosf/sinf returns) on all but double-rounding ties."""
    import ctypes as C
    oracle.vdo_oracle_sincos_exact.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    rng = np.random.default_rng(0)
    ang = np.concatenate([rng.uniform(0, 2 * np.pi, 20000), np.arange(0, 361, 1) * np.float32(np.pi / 180), [0.0, np.float32(2 * np.pi)]]).astype(np.float32)
    bad = 0
    for a in ang:
        s, c = C.c_float(), C.c_float()
        oracle.vdo_oracle_sincos_exact(float(a), C.byref(s), C.byref(c))
        es, ec = np.float32(np.sin(np.float64(a))), np.float32(np.cos(np.float64(a)))
        assert abs(s.value - float(es)) <= np.spacing(abs(es)) and abs(c.value - float(ec)) <= np.spacing(abs(ec)), a
        bad += (s.value != float(es)) + (c.value != float(ec))
    assert bad <= 2, bad


def test_brief_descriptor_against_a_brute_force_scalar_implementation(oracle):
    """computeOrbDescriptor (src/ORBextractor.cc:97-136) restated with numpy float32 scalars: same 32 bytes; the pattern
    table has the reference's shape (256 pairs inside the 31-px patch); rotation by 90 degrees moves the sample points as
    a rotation of the image does."""
    import ctypes as C
    oracle.vdo_oracle_orb_descriptor.argtypes = [R._u8(np.zeros(1, np.uint8)).__class__, C.c_int, C.c_float, C.c_float, C.c_float, R._u8(np.zeros(1, np.uint8)).__class__]
    oracle.vdo_oracle_orb_pattern.restype = C.POINTER(C.c_byte)
    pat = np.ctypeslib.as_array(oracle.vdo_oracle_orb_pattern(), (1024,)).astype(np.int32).reshape(256, 4)
    assert np.abs(pat).max() == 13 and (pat[:, 0] ** 2 + pat[:, 1] ** 2).max() <= 338 and np.unique(pat, axis=0).shape[0] == 256
    assert tuple(pat[0]) == (8, -3, 9, 5) and tuple(pat[255]) == (-1, -6, 0, -11)          # first / last entry of bit_pattern_31_
    rng = np.random.default_rng(3)
    w, h = 96, 80
    img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    f32 = np.float32
    factor = f32(np.pi / f32(180.0))
    for _ in range(40):
        px, py = f32(rng.integers(19, w - 19)), f32(rng.integers(19, h - 19))
        ang = f32(rng.uniform(0, 360))
        got = np.zeros(32, np.uint8)
        oracle.vdo_oracle_orb_descriptor(R._u8(img), w, px, py, ang, R._u8(got))
        rad = f32(ang * factor)
        a, b = f32(np.cos(np.float64(rad))), f32(np.sin(np.float64(rad)))
        exp = np.zeros(32, np.uint8)
        for i in range(256):
            x0, y0, x1, y1 = (f32(v) for v in pat[i])
            def val(x, y):
                r = int(np.rint(f32(f32(x * b) + f32(y * a)))); c = int(np.rint(f32(f32(x * a) - f32(y * b))))
                return int(img[int(py) + r, int(px) + c])
            exp[i // 8] |= (val(x0, y0) < val(x1, y1)) << (i % 8)
        assert np.array_equal(got, exp), (px, py, ang)
    # angle 0: unrotated pattern
    got = np.zeros(32, np.uint8)
    oracle.vdo_oracle_orb_descriptor(R._u8(img), w, 40.0, 40.0, 0.0, R._u8(got))
    exp = np.packbits([img[40 + p[1], 40 + p[0]] < img[40 + p[3], 40 + p[2]] for p in pat], bitorder="little")
    assert np.array_equal(got, exp)
    # angle 90 on the image == angle 0 on the image rotated by -90 degrees about the keypoint
    got90 = np.zeros(32, np.uint8)
    oracle.vdo_oracle_orb_descriptor(R._u8(img), w, 40.0, 40.0, 90.0, R._u8(got90))
    exp90 = np.packbits([img[40 + p[0], 40 - p[1]] < img[40 + p[2], 40 - p[3]] for p in pat], bitorder="little")
    assert np.array_equal(got90, exp90)


def test_extract_desc_keeps_the_keypoints_and_describes_on_the_blurred_level(oracle):
    gray = SF.make_gray(2, 400, 240)
    a, b = R.extract(oracle, gray), R.extract_desc(oracle, gray)
    for k in a:
        assert np.array_equal(a[k], b[k])
    import ctypes as C
    # first level-0 keypoint, recomputed by hand from the blurred level 0
    inner = R.pyramid(oracle, gray)[0][19:-19, 19:-19]
    blurred = R.blur7(oracle, inner)
    i = int(np.flatnonzero(b["octave"] == 0)[0])
    one = np.zeros(32, np.uint8)
    oracle.vdo_oracle_orb_descriptor.argtypes = [R._u8(one).__class__, C.c_int, C.c_float, C.c_float, C.c_float, R._u8(one).__class__]
    oracle.vdo_oracle_orb_descriptor(R._u8(blurred), blurred.shape[1], float(b["x"][i]), float(b["y"][i]), float(b["angle"][i]), R._u8(one))
    assert np.array_equal(one, b["desc"][i])
