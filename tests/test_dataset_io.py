"""Dataset IO of the reference's example driver (vdo_slam_amd/host/DatasetIO.{h,cc}: .flo, text instance masks, PNG) against files
written here with numpy / zlib.  Host only (no GPU)."""
import ctypes as C
import struct
import zlib

import numpy as np
import pytest

from vdo_slam_amd import _capi as K


@pytest.fixture(scope="module")
def host():
    L = K.load_host_lib()
    L.host_io_read_flo.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.c_void_p]
    L.host_io_load_mask.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p]
    L.host_io_read_png.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int), C.c_void_p]
    return L


def _png(path, arr, bit_depth, filters):
    """Minimal PNG writer: grey (h, w) or colour (h, w, 3|4), 8 or 16 bit, one filter type per row from `filters` (cycled)."""
    h, w = arr.shape[:2]
    ch = 1 if arr.ndim == 2 else arr.shape[2]
    ctype = {1: 0, 3: 2, 4: 6}[ch]
    raw = arr.astype(">u2" if bit_depth == 16 else np.uint8).reshape(h, -1).view(np.uint8).reshape(h, -1).astype(np.int32)
    bpp = ch * bit_depth // 8
    out = bytearray()
    prev = np.zeros(raw.shape[1], np.int32)
    for y in range(h):
        ft = filters[y % len(filters)]
        cur = raw[y]
        a = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]])
        c = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]])
        if ft == 0: pred = 0
        elif ft == 1: pred = a
        elif ft == 2: pred = prev
        elif ft == 3: pred = (a + prev) // 2
        else:
            p = a + prev - c
            pa, pb, pc = np.abs(p - a), np.abs(p - prev), np.abs(p - c)
            pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, prev, c))
        out.append(ft)
        out += ((cur - pred) & 255).astype(np.uint8).tobytes()
        prev = cur

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    z = zlib.compress(bytes(out), 6)
    data = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, bit_depth, ctype, 0, 0, 0))
    data += chunk(b"IDAT", z[: len(z) // 2]) + chunk(b"IDAT", z[len(z) // 2:]) + chunk(b"IEND", b"")      # two IDAT chunks
    open(path, "wb").write(data)


def test_read_flo(host, tmp_path):
    rng = np.random.default_rng(1)
    flow = rng.normal(0, 5, (37, 53, 2)).astype(np.float32)
    p = tmp_path / "a.flo"
    with open(p, "wb") as f:
        f.write(struct.pack("<fii", 202021.25, 53, 37)); f.write(flow.tobytes())
    dims = (C.c_int * 3)()
    assert host.host_io_read_flo(str(p).encode(), dims, None) == 0 and list(dims) == [37, 53, 2]
    out = np.zeros((37, 53, 2), np.float32)
    assert host.host_io_read_flo(str(p).encode(), dims, out.ctypes.data_as(C.c_void_p)) == 0
    assert np.array_equal(out, flow)
    bad = tmp_path / "b.flo"
    bad.write_bytes(b"not a flow file at all")
    assert host.host_io_read_flo(str(bad).encode(), dims, None) != 0
    assert host.host_io_read_flo(str(tmp_path / "missing.flo").encode(), dims, None) != 0


def test_load_mask_text(host, tmp_path):
    rng = np.random.default_rng(2)
    h, w = 40, 64
    m = np.zeros((h, w), np.int32)
    m[5:20, 10:30] = 3; m[25:35, 40:60] = 17; m[0, 0] = 101
    p = tmp_path / "m.txt"
    lines = [" ".join(str(v) for v in row) + " " for row in m]          # trailing blank like the dataset files
    lines.insert(12, "")                                                   # an empty line is skipped (`if(!s.empty())`)
    p.write_text("\n".join(lines) + "\n")
    out = np.full((h, w), -7, np.int32)
    assert host.host_io_load_mask(str(p).encode(), h, w, out.ctypes.data_as(C.c_void_p)) == 0
    assert np.array_equal(out, m)
    # a KITTI-sized mask parses in a few milliseconds (the reference's stringstream loop takes ~100x that)
    import time
    big = rng.integers(0, 4, (375, 1242)).astype(np.int32)
    pb = tmp_path / "big.txt"
    pb.write_text("\n".join(" ".join(map(str, row)) for row in big) + "\n")
    outb = np.zeros_like(big)
    t0 = time.perf_counter()
    assert host.host_io_load_mask(str(pb).encode(), 375, 1242, outb.ctypes.data_as(C.c_void_p)) == 0
    assert time.perf_counter() - t0 < 0.2 and np.array_equal(outb, big)


@pytest.mark.parametrize("bit_depth", [8, 16])
def test_read_png_grey_as_float(host, tmp_path, bit_depth):
    rng = np.random.default_rng(3)
    img = rng.integers(0, 2 ** bit_depth, (29, 41)).astype(np.uint16)
    img[10:20, 5:30] = np.arange(25)[None, :] * 7                           # smooth area: exercises the predictive filters
    p = tmp_path / "d.png"
    _png(p, img, bit_depth, filters=[0, 1, 2, 3, 4])
    dims = (C.c_int * 3)()
    assert host.host_io_read_png(str(p).encode(), 1, dims, None) == 0 and list(dims) == [29, 41, 1]
    out = np.zeros((29, 41), np.float32)
    assert host.host_io_read_png(str(p).encode(), 1, dims, out.ctypes.data_as(C.c_void_p)) == 0
    assert np.array_equal(out, img.astype(np.float32))


def test_read_png_colour_is_bgr(host, tmp_path):
    rng = np.random.default_rng(4)
    rgb = rng.integers(0, 256, (17, 23, 3)).astype(np.uint8)
    p = tmp_path / "c.png"
    _png(p, rgb, 8, filters=[4, 3, 1])
    dims = (C.c_int * 3)()
    out = np.zeros((17, 23, 3), np.uint8)
    assert host.host_io_read_png(str(p).encode(), 0, dims, out.ctypes.data_as(C.c_void_p)) == 0 and list(dims) == [17, 23, 3]
    assert np.array_equal(out, rgb[:, :, ::-1])
    assert host.host_io_read_png(str(tmp_path / "nope.png").encode(), 0, dims, None) != 0


def test_untrusted_headers_are_rejected_without_allocating(host, tmp_path):
    """A PNG whose IHDR claims 2^31 x 2^31 pixels, one whose claimed size cannot come out of its few compressed bytes, a .flo
    with absurd dimensions and a mask line with a 40-digit number: an error code, no allocation failure, no overflow."""
    import struct, zlib
    good = tmp_path / "g.png"
    _png(good, np.arange(12, dtype=np.uint8).reshape(3, 4), 8, filters=[0])
    raw = bytearray(good.read_bytes())
    dims = (C.c_int * 3)()
    for w, h in ((1 << 31, 1 << 31), (30000, 30000)):
        b = bytearray(raw)
        b[16:24] = struct.pack(">II", w & 0xffffffff, h & 0xffffffff)
        b[29:33] = struct.pack(">I", zlib.crc32(bytes(b[12:29])) & 0xffffffff)
        p = tmp_path / f"bad_{h}.png"
        p.write_bytes(bytes(b))
        assert host.host_io_read_png(str(p).encode(), 0, dims, None) != 0
    flo = tmp_path / "bad.flo"
    flo.write_bytes(struct.pack("<fii", 202021.25, 1 << 30, 1 << 30) + b"\0" * 64)
    assert host.host_io_read_flo(str(flo).encode(), dims, None) != 0
    m = tmp_path / "m.txt"
    m.write_text("1 " + "9" * 40 + " 3\n")
    out = np.zeros((1, 3), np.int32)
    assert host.host_io_load_mask(str(m).encode(), 1, 3, out.ctypes.data_as(C.c_void_p)) == 0
    assert out[0, 0] == 1 and out[0, 2] == 3 and out[0, 1] > 0
