"""CPU side of the image-preparation path: the numpy oracle (oracle/imgproc.py) against Pillow-made
golden vectors (tests/golden/r2_imgproc.npz, made by tests/golden/make_golden_imgproc.py) and against the installed
Pillow itself; the library's HOST-side plan function against the oracle's coefficients (no device work)."""
import ctypes as C
import os
import random

import numpy as np
import pytest

from oracle import imgproc as orc

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(HERE, "golden", "r2_imgproc.npz"))


def test_oracle_resample_vs_pillow_goldens(gold):
    n = 0
    for key in gold.files:
        parts = key.split("_")
        if len(parts) == 3 and "x" in parts[1] and parts[2] in ("f0", "f1"):
            oh, ow = (int(v) for v in parts[1].split("x"))
            got = orc.resample_lanczos(gold[parts[0] + "_in"], oh, ow, parts[2] == "f1")
            assert np.array_equal(got, gold[key]), key
            n += 1
    assert n == 14


def test_oracle_jitter_and_maps_vs_pillow_goldens(gold):
    for k in range(6):
        p = gold["jit%d_params" % k]
        got = orc.color_jitter(gold["jit_in"], [int(v) for v in p[:4]], p[4], p[5], p[6], int(p[7]))
        assert np.array_equal(got, gold["jit%d_out" % k]), k
    assert np.array_equal(orc.rgb2hsv(gold["cube"]), gold["cube_hsv"])
    assert np.array_equal(orc.to_L(gold["cube"]), gold["cube_L"])
    assert np.array_equal(orc.hsv2rgb(gold["cube"]), gold["cube_from_hsv"])


def test_oracle_vs_installed_pillow_exhaustive_maps():
    """all 2^24 triples through Convert.c's three maps; all 2^16 byte pairs through Blend.c at sampled factors."""
    Image = pytest.importorskip("PIL.Image")
    grid = np.stack(np.meshgrid(np.arange(256), np.arange(256), np.arange(256), indexing="ij"), -1)
    grid = grid.reshape(4096, 4096, 3).astype(np.uint8)
    im = Image.fromarray(grid)
    assert np.array_equal(orc.to_L(grid), np.asarray(im.convert("L")))
    assert np.array_equal(orc.rgb2hsv(grid), np.asarray(im.convert("HSV")))
    assert np.array_equal(orc.hsv2rgb(grid), np.asarray(Image.fromarray(grid, "HSV").convert("RGB")))
    pair = np.stack(np.meshgrid(np.arange(256), np.arange(256), indexing="ij"), -1).astype(np.uint8)
    A, B = np.repeat(pair[..., :1], 3, -1), np.repeat(pair[..., 1:], 3, -1)
    rng = np.random.default_rng(0)
    for alpha in list(rng.uniform(0.8, 1.2, 12)) + [0.0, 1.0, 0.5, 0.3, 1.0000001, 2.5, -0.25]:
        ref = np.asarray(Image.blend(Image.fromarray(A), Image.fromarray(B), float(alpha)))
        assert np.array_equal(orc.blend(A, B, alpha), ref), alpha


def test_oracle_vs_installed_pillow_kitti_sizes():
    """the four scales of both BASELINE resolutions from KITTI's frame sizes, with and without the flip."""
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(3)
    for (h, w) in ((375, 1242), (370, 1226)):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        for (H, W) in ((192, 640), (320, 1024)):
            for s in range(4):
                flip = bool(s % 2)
                im = Image.fromarray(img).transpose(Image.FLIP_LEFT_RIGHT) if flip else Image.fromarray(img)
                ref = np.asarray(im.resize((W >> s, H >> s), Image.LANCZOS))
                assert np.array_equal(orc.resample_lanczos(img, H >> s, W >> s, flip), ref), (h, w, H, s)


def test_oracle_jitter_vs_loader_color_jitter():
    """the CPU loader's Pillow chain (model_loader/kitti.py ColorJitter) == the oracle, 20 draws."""
    Image = pytest.importorskip("PIL.Image")
    from model_loader.kitti import ColorJitter
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (24, 80, 3), dtype=np.uint8)
    for seed in range(20):
        j = ColorJitter(random.Random(seed))
        ref = np.asarray(j(Image.fromarray(img)))
        assert np.array_equal(orc.color_jitter(img, j.order, j.b, j.c, j.s, int(j.h * 255)), ref), seed


def test_library_plan_function_equals_oracle_coefficients():
    """mdx_resample_plan is host code (Resample.c precompute_coeffs): callable without a GPU."""
    from mdx import _lib
    lib = _lib.lib()
    for (i, o) in ((1242, 640), (375, 192), (1242, 80), (375, 24), (1226, 320), (376, 320), (30, 60), (64, 64), (7, 1)):
        ks = lib.mdx_resample_ksize(i, o)
        ksize, bounds, kk = orc.resample_coeffs(i, o)
        assert ks == ksize
        b, k = np.zeros((o, 2), np.int32), np.zeros((ks, o), np.int32)          # the library's table is tap-major
        assert lib.mdx_resample_plan(i, o, b.ctypes.data_as(C.c_void_p), k.ctypes.data_as(C.c_void_p)) == 0
        assert np.array_equal(b, bounds) and np.array_equal(k.T, kk), (i, o)
        assert (k.sum(0) - (1 << 22)).__abs__().max() <= ks            # weights sum to one up to rounding
    assert lib.mdx_resample_ksize(0, 4) < 0 and lib.mdx_resample_plan(4, 4, None, None) < 0


def test_library_column_major_plan():
    """mdx_resample_plan_cols (include/mdx.h): the plan's weights column-major, zero-padded, forward and reversed -- what the rows
    form of the horizontal pass reads with scalar loads, two neighbouring columns over the union of their windows."""
    from mdx import _lib
    lib = _lib.lib()
    for (i, o) in ((1242, 640), (1242, 320), (1242, 160), (1242, 80), (1226, 320), (30, 60), (64, 64), (7, 1), (14000, 40)):
        ks, bounds, kk = orc.resample_coeffs(i, o)                        # kk [out][ksize]
        lead, row = C.c_int(-1), C.c_int(-1)
        assert lib.mdx_resample_plan_cols(i, o, C.byref(lead), C.byref(row), None) == 0
        lead, row = lead.value, row.value
        xmin, n = bounds[:, 0].astype(np.int64), bounds[:, 1].astype(np.int64)
        need = max([0] + list(np.diff(xmin)) + list(np.diff(xmin + n)))
        assert lead == need and row == lead + (ks + lead + 15) // 16 * 16 + 16
        tab = np.full((2, o, row), 12345, np.int32)
        l2, r2 = C.c_int(0), C.c_int(0)
        assert lib.mdx_resample_plan_cols(i, o, C.byref(l2), C.byref(r2), tab.ctypes.data_as(C.c_void_p)) == 0
        assert (l2.value, r2.value) == (lead, row)
        want = np.zeros_like(tab)
        for x in range(o):
            want[0, x, lead:lead + n[x]] = kk[x, :n[x]]
            want[1, x, lead:lead + n[x]] = kk[x, :n[x]][::-1]
        assert np.array_equal(tab, want), (i, o)
        # what the kernel relies on: a pair's union, read in chunks of 16 from either row (the second at -d), stays inside a row
        for x in range(o - 1):
            d = int(xmin[x + 1] - xmin[x])
            U = max(int(n[x]), d + int(n[x + 1]))
            assert 0 <= lead - d and lead + (U + 15) // 16 * 16 <= row
    assert lib.mdx_resample_plan_cols(0, 4, C.byref(C.c_int()), C.byref(C.c_int()), None) < 0
    assert lib.mdx_resample_plan_cols(4, 4, None, None, None) < 0


def test_unit_from_byte_sequence_is_the_ieee_quotient():
    """csrc/imgproc.hip unit_from_byte: q0 = b * rc, r = fma(-255, q0, b), q = fma(r, rc, q0) with rc = fl32(1 / 255) equals
    float32(b) / float32(255) for every byte -- checked in exact rational arithmetic (each fma rounds once)."""
    from fractions import Fraction

    def rnd(x):
        g = np.float32(float(x))
        cands = [np.nextafter(g, np.float32(-np.inf)), g, np.nextafter(g, np.float32(np.inf))]
        return min(cands, key=lambda c: (abs(Fraction(float(c)) - x), int(c.view(np.uint32)) & 1))
    rc = np.float32(1.0) / np.float32(255.0)
    assert int(rc.view(np.uint32)) == 0x3B808081
    frc = Fraction(float(rc))
    for b in range(256):
        q0 = rnd(Fraction(b) * frc)
        r = rnd(Fraction(b) - 255 * Fraction(float(q0)))
        q = rnd(Fraction(float(q0)) + Fraction(float(r)) * frc)
        ref = np.float32(b) / np.float32(255.0)
        assert int(q.view(np.uint32)) == int(ref.view(np.uint32)), b
