"""The reference's only fixtures - example/kitti-0000-0013.yaml, kitti-0018-0020.yaml, omd.yaml - through the PRODUCT's settings reader
(host/System.cc read_tracking_settings = what Tracking::Tracking reads, /root/reference/src/Tracking.cc:53-161): every key the
reference reads, with cv::FileNode's conversions (float / int casts, missing key = 0, fps 0 -> 30).  Host-only code: runs without a GPU.
The expected values are literals taken from the three files; when the checkout is present the files themselves are parsed (by the
product and, independently, by a few lines of Python here), otherwise copies of their key/value lines written from the literals are."""
import ctypes as C
import os

import numpy as np
import pytest

from vdo_slam_amd import _capi as K

REF_EXAMPLE = os.path.join(os.environ.get("VDO_REFERENCE_ROOT", "/root/reference"), "example")

# declaration order of TrackingSettings (host/System.cc) = reading order of src/Tracking.cc:53-161 (+ width / height)
KEYS = ["Camera.fx", "Camera.fy", "Camera.cx", "Camera.cy", "Camera.k1", "Camera.k2", "Camera.p1", "Camera.p2", "Camera.k3", "Camera.bf", "Camera.fps",
        "Camera.RGB", "ORBextractor.nFeatures", "ORBextractor.scaleFactor", "ORBextractor.nLevels", "ORBextractor.iniThFAST", "ORBextractor.minThFAST",
        "ChooseData", "ThDepthBG", "ThDepthOBJ", "DepthMapFactor", "MaxTrackPointBG", "MaxTrackPointOBJ", "SFMgThres", "SFDsThres", "WINDOW_SIZE",
        "OVERLAP_SIZE", "UseSampleFeature", "Camera.width", "Camera.height"]
INT_KEYS = {"Camera.RGB", "ORBextractor.nFeatures", "ORBextractor.nLevels", "ORBextractor.iniThFAST", "ORBextractor.minThFAST", "ChooseData", "MaxTrackPointBG",
            "MaxTrackPointOBJ", "WINDOW_SIZE", "OVERLAP_SIZE", "UseSampleFeature", "Camera.width", "Camera.height"}

_KITTI = {"Camera.fx": 721.5377, "Camera.fy": 721.5377, "Camera.cx": 609.5593, "Camera.cy": 172.8540, "Camera.k1": 0.0, "Camera.k2": 0.0, "Camera.p1": 0.0,
          "Camera.p2": 0.0, "Camera.k3": 0.0, "Camera.width": 1242, "Camera.height": 375, "Camera.fps": 10.0, "Camera.bf": 387.5744, "Camera.RGB": 1,
          "ChooseData": 2, "DepthMapFactor": 256.0, "ThDepthBG": 40.0, "ThDepthOBJ": 25.0, "MaxTrackPointBG": 1200, "MaxTrackPointOBJ": 800, "SFMgThres": 0.12,
          "SFDsThres": 0.3, "WINDOW_SIZE": 20, "OVERLAP_SIZE": 4, "UseSampleFeature": 0, "ORBextractor.nFeatures": 2500, "ORBextractor.scaleFactor": 1.2,
          "ORBextractor.nLevels": 8, "ORBextractor.iniThFAST": 20, "ORBextractor.minThFAST": 7}
EXPECTED = {
    "kitti-0000-0013.yaml": dict(_KITTI),
    # no Camera.k3 line in this file: cv::FileNode of a missing key converts to 0
    "kitti-0018-0020.yaml": {**{k: v for k, v in _KITTI.items() if k != "Camera.k3"}, "Camera.fx": 718.8560, "Camera.fy": 718.8560, "Camera.cx": 607.1928,
                             "Camera.cy": 185.2157, "Camera.bf": 388.1822},
    "omd.yaml": {**_KITTI, "Camera.fx": 618.3587036132812, "Camera.fy": 618.5924072265625, "Camera.cx": 328.9866333007812, "Camera.cy": 237.7507629394531,
                 "Camera.width": 640, "Camera.height": 480, "Camera.fps": 30.0, "ChooseData": 1, "DepthMapFactor": 1000.0, "SFMgThres": 0.02, "SFDsThres": 0.99,
                 "UseSampleFeature": 1, "ORBextractor.nFeatures": 3000},
}


def product_read(path):
    L = K.load_host_lib()
    L.host_settings_read.argtypes = [C.c_char_p, K.c_double_p]
    out = np.zeros(31)
    rc = L.host_settings_read(str(path).encode(), out.ctypes.data_as(K.c_double_p))
    return rc, dict(zip(KEYS, out[:30]))


def python_read(path):
    """independent of the product: "key: value [# comment]" lines of a YAML 1.0 file"""
    out = {}
    for line in open(path):
        line = line.split("#", 1)[0].strip()
        if not line or line.startswith("%") or ":" not in line:
            continue
        k, v = line.split(":", 1)
        try:
            out[k.strip()] = float(v)
        except ValueError:
            pass
    return out


def as_reference_reads(k, v):
    """cv::FileNode -> int truncates a real, -> float rounds a double once"""
    return float(int(v)) if k in INT_KEYS else float(np.float32(v))


@pytest.mark.parametrize("name", sorted(EXPECTED))
def test_reference_example_settings_through_the_product_reader(name, tmp_path):
    want = EXPECTED[name]
    path = os.path.join(REF_EXAMPLE, name)
    if os.path.exists(path):
        raw = python_read(path)
        assert raw == {k: float(v) for k, v in want.items()}, "the literals of this test no longer describe the reference's file"
    else:   # GPU box / no checkout: the same key/value lines, in the reference's layout (comment lines, trailing comments)
        path = tmp_path / name
        path.write_text("%YAML:1.0\n\n# Camera calibration\n" + "".join(f"{k}: {v!r}   # was {v}\n" if i % 3 == 0 else f"{k}: {v!r}\n" for i, (k, v) in enumerate(want.items())))
    rc, got = product_read(path)
    assert rc == 0
    for k in KEYS:
        exp = as_reference_reads(k, want.get(k, 0.0))
        assert got[k] == exp, (k, got[k], exp)
    assert set(want) <= set(KEYS), "a key of the file is not read by the product"


def test_missing_file_and_fps_default(tmp_path):
    rc, _ = product_read(tmp_path / "nope.yaml")
    assert rc == -1                                           # Tracking::Tracking then exits(-1) like the reference (src/System.cc:35-39)
    p = tmp_path / "s.yaml"
    p.write_text("%YAML:1.0\nCamera.fx: 500\nCamera.fps: 0\n")
    rc, got = product_read(p)
    assert rc == 0 and got["Camera.fps"] == 30.0 and got["Camera.fx"] == 500.0 and got["Camera.RGB"] == 0.0   # src/Tracking.cc:81-83
