"""GPU parity of csrc/imgproc.hip (the KITTI loaders' Pillow resizes, colour jitter and ToTensor):
bit-exact against the numpy oracle (oracle/imgproc.py) and against the installed Pillow itself."""
import random

import numpy as np
import pytest
import torch

from oracle import imgproc as orc

pytestmark = pytest.mark.gpu
PIL = pytest.importorskip("PIL")
from PIL import Image  # noqa: E402


@pytest.fixture(scope="module")
def G():
    import gpu_util
    return gpu_util


@pytest.fixture(scope="module")
def IP(G):
    from mdx import imgproc
    return imgproc


def _natural(rng, h, w):
    """smooth structure + noise, uint8 [h,w,3] (white noise alone never exercises the clipping of Lanczos overshoot)"""
    lo = rng.random((max(h // 8, 2), max(w // 8, 2), 3))
    img = np.asarray(Image.fromarray((lo * 255).astype(np.uint8)).resize((w, h), Image.BICUBIC)).astype(np.int32)
    img = img + rng.integers(-12, 13, img.shape)
    img[: h // 4, : w // 4] = rng.integers(0, 2, (h // 4, w // 4, 1)) * 255        # hard edges: overshoot -> clip8
    return img.clip(0, 255).astype(np.uint8)


def _stack(imgs):
    hmax, wmax = max(i.shape[0] for i in imgs), max(i.shape[1] for i in imgs)
    out = np.zeros((len(imgs), hmax, wmax, 3), np.uint8)
    for n, i in enumerate(imgs):
        out[n, : i.shape[0], : i.shape[1]] = i
    return out


def _pil_resize(img, oh, ow, flip):
    im = Image.fromarray(img)
    if flip:
        im = im.transpose(Image.FLIP_LEFT_RIGHT)
    return np.asarray(im.resize((ow, oh), Image.LANCZOS))


@pytest.mark.parametrize("case", [
    dict(sizes=[(375, 1242)] * 3, out=(192, 640)),
    dict(sizes=[(375, 1242), (370, 1226), (376, 1241), (374, 1238)], out=(96, 320)),       # KITTI's sizes, one batch
    dict(sizes=[(375, 1242), (370, 1226)], out=(24, 80)),                                  # 95 taps
    dict(sizes=[(375, 1242), (376, 1241)], out=(320, 1024)),                               # BASELINE configs[3]
    dict(sizes=[(20, 30), (17, 30)], out=(40, 60)),                                        # upscale
    dict(sizes=[(48, 64)], out=(48, 64)),                                                  # identity plan
    dict(sizes=[(50, 50)], out=(50, 25)),
    dict(sizes=[(5, 2), (4, 3), (1, 2)], out=(3, 2)),                                      # narrowest sources, odd rows
    dict(sizes=[(33, 47)] * 75, out=(16, 24)),                                             # > MDX_IMG_JOBS jobs: two launches
    dict(sizes=[(20 + (i % 3), 40 + i) for i in range(19)], out=(8, 16)),                  # 19 plans per axis > 16 per launch
    dict(sizes=[(9, 4200), (7, 4100)], out=(8, 2112)),                                     # vertical pass: a row wider than a block (264 threads)
    dict(sizes=[(20, 30), (19, 28)], out=(130, 136)),                                      # upscale large enough for the 8-byte vertical form, 15 rows per block
    dict(sizes=[(40, 100)], out=(30, 36)),                                                 # out_w % 4 == 0, % 8 != 0: dword stores, byte vertical pass
    dict(sizes=[(40, 300), (37, 290)], out=(129, 136)),                                    # 8-byte vertical form (two rows per thread) with an ODD number of rows: a lone last row
    dict(sizes=[(300, 40)], out=(67, 256)),                                                # the same, downscaling rows (windows of neighbouring rows 4.5 apart), odd rows
])
def test_resize_lanczos_bit_exact(G, IP, case):
    rng = np.random.default_rng(len(case["sizes"]) * 1000 + case["out"][1])
    imgs = [_natural(rng, h, w) for (h, w) in case["sizes"]]
    flips = [bool(n % 2) for n in range(len(imgs))]
    oh, ow = case["out"]
    plans = IP.plan_cache("cuda:0")
    u8, f32 = IP.resize_lanczos(plans, torch.from_numpy(_stack(imgs)).cuda(), case["sizes"], flips, (oh, ow), want_u8=True)
    u8, f32 = u8.cpu().numpy(), f32.cpu().numpy()
    for n, img in enumerate(imgs):
        ref = orc.resample_lanczos(img, oh, ow, flips[n])
        assert np.array_equal(ref, _pil_resize(img, oh, ow, flips[n]))                      # oracle == Pillow
        assert np.array_equal(u8[n], ref.transpose(2, 0, 1)), "image %d" % n
        assert np.array_equal(f32[n].view(np.uint32), orc.to_tensor(ref).view(np.uint32)), "ToTensor %d" % n


def test_resize_filter_widths(G, IP):
    """the rows form of the horizontal pass over the filter widths of the KITTI pyramid (25, 49, 95 taps) and around its
    tap-group boundaries (33, 65, 97, 129 taps: one and several 64-tap groups, blocks of 16 ... 4 columns), with flips."""
    rng = np.random.default_rng(78)
    plans = IP.plan_cache("cuda:0")
    cases = [((375, 1242), (96, 320)), ((375, 1242), (48, 160)), ((375, 1242), (24, 80)), ((21, 1226), (21, 230)),
             ((9, 1000), (9, 94)), ((9, 1000), (7, 63)), ((5, 1300), (5, 61)), ((70, 900), (33, 17))]
    for (h, w), out in cases:
        imgs = [_natural(rng, h, w), _natural(rng, h, w - 5)]
        sizes, flips = [(h, w), (h, w - 5)], [False, True]
        src = torch.from_numpy(_stack(imgs)).cuda()
        a = IP.resize_lanczos(plans, src, sizes, flips, out, want_u8=True)[0]
        for n in range(2):
            assert np.array_equal(a[n].cpu().numpy(), orc.resample_lanczos(imgs[n], out[0], out[1], flips[n]).transpose(2, 0, 1)), ((h, w), out, n)


def test_resize_rows_form_equals_gather_form(G, IP):
    """the horizontal pass's two forms on the same jobs: rows form (two columns per pass over the union of their windows,
    weights by scalar loads from the column-major table) against the gather form (job without the table) -- byte for byte,
    KITTI pyramid shapes and ragged ones, flipped and not, odd column counts (a lone last column per wave)."""
    rng = np.random.default_rng(79)
    rows, gather = IP.plan_cache("cuda:0"), IP.plan_cache("cuda:0", cols=False)
    for (h, w), out in [((70, 1242), (36, 640)), ((70, 1242), (18, 320)), ((70, 1242), (9, 160)), ((70, 1242), (5, 80)),
                        ((70, 1241), (11, 637)), ((33, 500), (20, 250)), ((33, 301), (20, 299)), ((12, 64), (12, 64)), ((9, 40), (9, 90))]:
        imgs = [_natural(rng, h, w), _natural(rng, h - 3, w - 7)]
        sizes, flips = [(h, w), (h - 3, w - 7)], [False, True]
        src = torch.from_numpy(_stack(imgs)).cuda()
        a = IP.resize_lanczos(rows, src, sizes, flips, out, want_u8=True)[0]
        b = IP.resize_lanczos(gather, src, sizes, flips, out, want_u8=True)[0]
        assert torch.equal(a, b), ((h, w), out)
        for n in range(2):
            assert np.array_equal(a[n].cpu().numpy(), orc.resample_lanczos(imgs[n], out[0], out[1], flips[n]).transpose(2, 0, 1)), ((h, w), out, n)


def test_resize_reads_nothing_behind_an_image(G, IP):
    """The bytes beside and below an image in its slot -- and behind the last pixel of its last row (the rows form
    treated in_h * in_stride bytes as readable, a crop view of a larger image has nothing there) -- are not part of the input:
    the same frames in slots padded with 0 and with 255 give the same bytes, both passes' forms, flipped and not."""
    rng = np.random.default_rng(83)
    for (h, w), out in [((70, 1242), (36, 640)), ((33, 301), (20, 299)), ((9, 40), (9, 90)), ((17, 23), (5, 7))]:
        imgs = [_natural(rng, h, w), _natural(rng, h - 3, w - 7)]
        sizes, flips = [(h, w), (h - 3, w - 7)], [False, True]
        a0 = _stack(imgs)
        a1 = np.full_like(a0, 255)
        for n, i in enumerate(imgs):
            a1[n, : i.shape[0], : i.shape[1]] = i
        for plans in (IP.plan_cache("cuda:0"), IP.plan_cache("cuda:0", cols=False)):
            r0 = IP.resize_lanczos(plans, torch.from_numpy(a0).cuda(), sizes, flips, out, want_u8=True)[0]
            r1 = IP.resize_lanczos(plans, torch.from_numpy(a1).cuda(), sizes, flips, out, want_u8=True)[0]
            assert torch.equal(r0, r1), ((h, w), out)


def test_resize_wide_source(G, IP):
    """a source 350x wider than its output (2101 taps per column): beyond the rows form's LDS tile, the gather form."""
    rng = np.random.default_rng(77)
    img = _natural(rng, 9, 14000)
    got = IP.resize_lanczos(IP.plan_cache("cuda:0"), torch.from_numpy(img[None]).cuda(), [(9, 14000)], [True], (5, 40), want_u8=True)[0]
    assert np.array_equal(got[0].cpu().numpy(), orc.resample_lanczos(img, 5, 40, True).transpose(2, 0, 1))


def test_color_maps_exhaustive(G, IP):
    """every RGB triple through RGB->HSV and RGB->L, every HSV triple through HSV->RGB: equal to Pillow's convert()."""
    grid = np.stack(np.meshgrid(np.arange(256), np.arange(256), np.arange(256), indexing="ij"), -1)
    grid = grid.reshape(4096, 4096, 3).astype(np.uint8)
    planar = torch.from_numpy(np.ascontiguousarray(grid.reshape(-1, 3).T)).cuda()
    im = Image.fromarray(grid)
    for mode, ref in (("hsv", np.asarray(im.convert("HSV"))), ("rgb", np.asarray(Image.fromarray(grid, "HSV").convert("RGB")))):
        got = IP.color_convert(planar, mode).cpu().numpy().T.reshape(4096, 4096, 3)
        assert np.array_equal(got, ref), mode
    got = IP.color_convert(planar, "L").cpu().numpy().reshape(4096, 4096)
    assert np.array_equal(got, np.asarray(im.convert("L")))
    sub = np.ascontiguousarray(grid[::37, ::41])
    assert np.array_equal(orc.rgb2hsv(sub), np.asarray(Image.fromarray(sub).convert("HSV")))


def test_color_jitter_bit_exact(G, IP):
    """30 draws of (order, factors) in the loader's ranges + factors inside [0,1] (truncation branch) and partial
    orders; against the oracle and the loader's Pillow ColorJitter."""
    from model_loader.kitti import ColorJitter
    rng = np.random.default_rng(5)
    imgs = [_natural(rng, 48, 96) for _ in range(34)]
    params, refs = [], []
    for n, img in enumerate(imgs):
        if n == 7:
            params.append(None)
            refs.append(None)
            continue
        j = ColorJitter(random.Random(n))
        if n >= 30:
            j.b, j.c, j.s = 0.3 + 0.1 * (n - 30), 0.9, 0.55
            j.order = j.order[: 2 + (n - 30) % 3]
        p = (j.order, j.b, j.c, j.s, int(j.h * 255))
        params.append(p)
        ref = orc.color_jitter(img, *p)
        assert np.array_equal(ref, np.asarray(j(Image.fromarray(img))))                      # oracle == Pillow chain
        refs.append(ref)
    src = torch.from_numpy(np.stack([i.transpose(2, 0, 1) for i in imgs])).contiguous().cuda()
    out = torch.full((len(imgs), 3, 48, 96), -1.0, device="cuda")
    IP.color_jitter(src, params, out)
    out = out.cpu().numpy()
    assert (out[7] == -1.0).all()                                                            # untouched without params
    for n, ref in enumerate(refs):
        if ref is not None:
            assert np.array_equal(out[n].view(np.uint32), orc.to_tensor(ref).view(np.uint32)), "image %d" % n


def test_color_jitter_odd_sizes_and_empty_chains(G, IP):
    """image sizes that are not multiples of four (the kernels' pixel-by-pixel tails; a plane that starts at an odd address
    takes them for the whole image), chains with Contrast first / last, and the empty chain (a plain ToTensor)."""
    from model_loader.kitti import ColorJitter
    rng = np.random.default_rng(15)
    for (h, w) in ((7, 9), (5, 13), (6, 10)):
        imgs = [_natural(rng, h, w) for _ in range(9)]
        params, refs = [], []
        for n, img in enumerate(imgs):
            if n == 8:
                p = ([4, 4, 4, 4], 1.0, 1.0, 1.0, 0)                                    # the empty chain
                ref = img
            else:
                j = ColorJitter(random.Random(100 + n))
                order = list(j.order)
                if n < 2:
                    order.remove(1); order.insert(0, 1)                                 # Contrast first
                elif n < 4:
                    order.remove(1); order.append(1)                                    # Contrast last
                p = (order, j.b, j.c, j.s, int(j.h * 255))
                ref = orc.color_jitter(img, *p)
            params.append(p)
            refs.append(ref)
        src = torch.from_numpy(np.stack([i.transpose(2, 0, 1) for i in imgs])).contiguous().cuda()
        out = IP.color_jitter(src, params).cpu().numpy()
        for n, ref in enumerate(refs):
            assert np.array_equal(out[n].view(np.uint32), orc.to_tensor(ref).view(np.uint32)), ((h, w), n)


@pytest.mark.parametrize("hw", [(192, 640), (320, 1024)])
def test_image_prep_equals_cpu_loader_arithmetic(G, IP, hw):
    """the batch-level stage against what the CPU loader computes per sample with Pillow (model_loader/kitti.py), at
    both BASELINE resolutions."""
    H, W = hw
    from model_loader.kitti import ColorJitter, to_tensor
    rng = np.random.default_rng(9)
    B, frames, sizes = 3, [0, -1, 1], [(375, 1242), (370, 1226), (376, 1241)]
    raw = {f: [_natural(rng, h, w) for (h, w) in sizes] for f in frames}
    flips = [False, True, False]
    jit = [ColorJitter(random.Random(3)), None, ColorJitter(random.Random(4))]
    rows = [[0.0] * 9 if j is None else [1.0] + list(j.order) + [j.b, j.c, j.s, int(j.h * 255)] for j in jit]
    batch = {("raw", f): torch.from_numpy(_stack(raw[f])) for f in frames}
    batch.update({"raw_size": torch.tensor(sizes, dtype=torch.int32), "raw_flip": torch.tensor(flips),
                  "raw_jitter": torch.tensor(rows, dtype=torch.float64), ("K", 0): torch.eye(4).repeat(B, 1, 1)})
    prep = IP.image_prep(H, W, frames, 4, "cuda:0")
    out = prep(batch)
    assert ("raw", 0) not in out and "raw_size" not in out and ("K", 0) in out
    for f in frames:
        for b in range(B):
            im = Image.fromarray(raw[f][b])
            if flips[b]:
                im = im.transpose(Image.FLIP_LEFT_RIGHT)
            for s in range(4 if f == 0 else 1):
                small = im.resize((W >> s, H >> s), Image.LANCZOS)
                assert torch.equal(out[("color", f, s)][b].cpu(), to_tensor(small)), (f, b, s)
                if s == 0:
                    aug = to_tensor(jit[b](small)) if jit[b] is not None else to_tensor(small)
                    assert torch.equal(out[("color_aug", f, 0)][b].cpu(), aug), (f, b)
    assert ("color", -1, 1) not in out and ("color_aug", 0, 1) not in out


def test_to_tensor_divides_like_totensor(G, IP):
    """ToTensor (kitti_mono.py:283,351) is float32(x) / 255 with a true division.  mdx_to_tensor_u8 reproduces it for
    every byte value at every alignment; a multiplication by 1/255 (what torch's GPU `x / 255.0` does with a Python
    scalar) would be one ulp off for 126 of the 256 values -- the uint8 loader path of round 2 carried that error."""
    ref = torch.arange(256, dtype=torch.uint8).float().div(255)                 # ATen CPU: IEEE division
    assert (ref.numpy() != (np.arange(256, dtype=np.float32) * (np.float32(1) / np.float32(255)))).sum() == 126
    rng = np.random.default_rng(0)
    for n in (1, 15, 16, 17, 255, 4096, 4097, 3 * 192 * 640 + 5):
        x = torch.from_numpy(rng.integers(0, 256, n, dtype=np.uint8))
        for off in (0, 1, 3):                                                    # unaligned sources take the byte path
            src = torch.zeros(n + off, dtype=torch.uint8, device="cuda:0")
            src[off:] = x.cuda()
            out = IP.to_tensor(src[off:])
            assert torch.equal(out.cpu(), ref[x.long()]), (n, off)
    img = torch.from_numpy(rng.integers(0, 256, (2, 3, 5, 7), dtype=np.uint8)).cuda()
    assert IP.to_tensor(img).shape == img.shape and torch.equal(IP.to_tensor(img).cpu(), ref[img.cpu().long()])


def test_imgproc_refuses_bad_input(G, IP):
    from mdx._lib import MdxError
    plans = IP.plan_cache("cuda:0")
    with pytest.raises(MdxError):
        IP.resize_lanczos(plans, torch.zeros(1, 8, 8, 3, dtype=torch.uint8), [(8, 8)], [False], (4, 4))     # CPU tensor
    with pytest.raises(MdxError):
        IP.resize_lanczos(plans, torch.zeros(1, 8, 8, 3, dtype=torch.uint8).cuda(), [(9, 8)], [False], (4, 4))
    with pytest.raises(MdxError):
        IP.color_jitter(torch.zeros(1, 3, 8, 8, dtype=torch.uint8).cuda(), [([0, 0, 1, 2], 1.0, 1.0, 1.0, 0)])


def test_kitti_loader_gpu_prep_equals_pillow_path(G, IP, tmp_path):
    """End to end on a JPEG tree: KITTIDataset(gpu_prep) -> padded collate -> image_prep  ==  the Pillow loader's
    entries for the same random draws (flip and jitter included), every entry a step reads."""
    import fake_kitti
    from model_loader import KITTIMonoStereoDataset
    from model_loader.kitti import collate_raw
    from model_tool.processor import step_reads
    names = fake_kitti.make(str(tmp_path), n_frames=6)
    frames = [0, -1, 1, "s"]
    cpu = KITTIMonoStereoDataset(str(tmp_path), names, True, frames, 192, 640, "jpg", 4)
    raw = KITTIMonoStereoDataset(str(tmp_path), names, True, frames, 192, 640, "jpg", 4)
    raw.gpu_prep = True
    ref, samples = [], []
    for i in range(len(names)):
        random.seed(40 + i)
        ref.append(cpu[i])
        random.seed(40 + i)
        samples.append(raw[i])
    assert any(bool(s["raw_flip"]) for s in samples) and any(bool(s["raw_jitter"][0]) for s in samples)
    out = IP.image_prep(192, 640, frames, 4, "cuda:0")(collate_raw(samples, step_reads))
    for key, value in out.items():
        if isinstance(key, tuple) and key[0] in ("color", "color_aug"):
            for b in range(len(names)):
                assert torch.equal(value[b].cpu(), ref[b][key]), (key, b)
    for b in range(len(names)):
        assert torch.equal(out["stereo"][b], ref[b]["stereo"])
        assert torch.equal(out[("depth", 0)][b].cpu(), ref[b][("depth", 0)])          # scattered from the sparse pairs
    wanted = {k for k in ref[0] if step_reads(k)}
    assert wanted == set(out.keys()), wanted ^ set(out.keys())
