"""Parity of the HIP blob path (through the C-ABI) with the CPU oracle: bit-exact masks, contours, centroids."""
import os

import numpy as np
import pytest

import oracle
from mocapv2_amd.synth import MILD_DIST, ZERO_DIST, Scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


def make_ctx(W, H, K=None, dist=None, n_slots=1):
    from mocapv2_amd.engine import MocapContext
    ctx = MocapContext(W, H, n_slots)
    K = np.array([[0.7 * W, 0, W / 2.0], [0, 0.7 * W, H / 2.0], [0, 0, 1]]) if K is None else K
    ident = ctx.set_undistort(0, K, np.zeros(5) if dist is None else dist)
    return ctx, K, ident


def rand_frames(rng, n, H, W, bright=0.35, blobs=3):
    img = rng.integers(0, 200, (n, H, W), dtype=np.uint8)
    img[rng.random((n, H, W)) < bright] = 255
    yy, xx = np.mgrid[0:H, 0:W]
    for i in range(n):
        for _ in range(blobs):
            cx, cy, r = rng.uniform(-5, W + 5), rng.uniform(-5, H + 5), rng.uniform(3, max(4, min(H, W) / 4))
            img[i][(xx - cx) ** 2 + (yy - cy) ** 2 <= r * r] = 255
    return img


SIZES = [(1, 1), (5, 3), (7, 9), (31, 17), (33, 40), (64, 64), (239, 20), (240, 33), (241, 50), (250, 130), (257, 140),
         (481, 31), (500, 300), (960, 540)]


@pytest.mark.parametrize("W,H", SIZES)
def test_filter_mask_identity(torch_cuda, W, H):
    from gpu_util import unpack_mask
    torch = torch_cuda
    rng = np.random.default_rng(W * 1000 + H)
    frames = rand_frames(rng, 3, H, W)
    ctx, K, ident = make_ctx(W, H)
    assert ident
    mask = ctx.filter_mask(torch.from_numpy(frames).cuda())
    got, pad = unpack_mask(mask, W)
    assert not pad.any()
    for i in range(3):
        exp = oracle.image_filter(frames[i], 0) != 0
        assert np.array_equal(got[i], exp), f"image {i}: {np.argwhere(got[i] != exp)[:5]}"


@pytest.mark.parametrize("W,H", [(97, 61), (640, 360), (1920, 1080)])
def test_filter_mask_against_scipy_without_the_c_oracle(torch_cuda, W, H):
    """The filter chain of image_filter_gpu (lib/ImageOperations.py:23-31) for an identity lens, restated with SciPy in the test
    itself -- no line shared with the kernels or with oracle/blob_oracle.c: numba's blur = integer sum of the in-bounds 5x5 taps
    floor-divided by their count (lib/CudaOperations.py:5-22), cv.threshold(216.75) = > 216, cv.medianBlur(5) = median with
    replicated borders -- against mocap_filter_mask, bit for bit."""
    import scipy.ndimage as ndi
    from gpu_util import unpack_mask
    torch = torch_cuda
    rng = np.random.default_rng(W * 7 + H)
    frames = dark_frames(rng, 2, H, W, n_discs=6 if W > 200 else 2, salt=0.002, noise_max=90)
    frames[1, : H // 3] = np.maximum(frames[1, : H // 3], rng.integers(150, 256, (H // 3, W), dtype=np.uint8))  # a bright band
    ctx, K, ident = make_ctx(W, H)
    assert ident
    got, pad = unpack_mask(ctx.filter_mask(torch.from_numpy(frames).cuda()), W)
    assert not pad.any()
    ones = np.ones((5, 5), np.int64)
    taps = ndi.convolve(np.ones((H, W), np.int64), ones, mode="constant", cval=0)
    for i in range(2):
        S = ndi.convolve(frames[i].astype(np.int64), ones, mode="constant", cval=0)
        thr = ((S // taps) > 216).astype(np.uint8) * 255
        exp = ndi.median_filter(thr, size=5, mode="nearest") != 0
        assert np.array_equal(got[i], exp), (i, np.argwhere(got[i] != exp)[:5])
        assert exp.any()


def test_borders_against_scipy_labels_and_picks_theorem(torch_cuda):
    """Oracle-free properties of the contour stage (SURVEY.md 8c lists them as stand-ins for the missing cv2 pair), on the GPU's own
    mask of 1080p frames with 24 discs: as many outer borders as 8-connected foreground components and as many hole borders as
    enclosed 4-connected background components (scipy.ndimage.label); every border's polygon area obeys Pick's theorem with the
    lattice points the walk visits -- outer border of a hole-free component: pixels - steps / 2 - 1 -- and the truncated centroid of
    a disc lies within a pixel of its pixel centroid."""
    import scipy.ndimage as ndi
    from gpu_util import unpack_mask
    torch = torch_cuda
    W, H = 1920, 1080
    rng = np.random.default_rng(77)
    frames = dark_frames(rng, 2, H, W, n_discs=24, salt=0.001)
    ctx, K, ident = make_ctx(W, H)
    ctx.set_blob_params(min_area=50.0, min_circ=0.3)
    mask_dev = ctx.filter_mask(torch.from_numpy(frames).cuda())
    masks, _ = unpack_mask(mask_dev, W)
    xy, cnt, recs = ctx.contours_from_mask(mask_dev, max_blobs=128, debug_cap=384)
    eight = np.ones((3, 3), int)
    checked = 0
    for i in range(2):
        m = masks[i]
        lab, n_fg = ndi.label(m, structure=eight)
        bg, n_bg = ndi.label(~np.pad(m, 1))  # 4-connected background, the frame's margin joins everything outside
        outer = [r for r in recs[i] if not r["is_hole"]]
        holes = [r for r in recs[i] if r["is_hole"]]
        assert len(outer) == n_fg and len(holes) == n_bg - 1, (len(outer), n_fg, len(holes), n_bg)
        sizes = ndi.sum(m, lab, index=np.arange(1, n_fg + 1))
        cents = ndi.center_of_mass(m, lab, index=np.arange(1, n_fg + 1))
        filled = ndi.binary_fill_holes(m)
        for r in outer:
            k = lab[r["sy"], r["sx"]]
            assert k > 0
            comp = lab == k
            if (filled & ~m)[ndi.binary_dilation(comp, structure=eight)].any():
                continue  # a component with a hole: its outer polygon encloses the hole's pixels too
            if r["steps"] == 0:
                continue
            assert r["a00"] < 0 or r["area"] == 0  # outer borders run this way round
            if r["area"] == sizes[k - 1] - r["steps"] / 2 - 1:  # Pick: interior + boundary / 2 - 1 (no pixel visited twice)
                checked += 1
                if r["kept"]:
                    assert abs(r["cx"] + 0.5 - cents[k - 1][1]) <= 1.0 and abs(r["cy"] + 0.5 - cents[k - 1][0]) <= 1.0
            else:  # a pixel of a one-pixel-wide part is visited twice: the polygon's lattice points are fewer than the steps
                assert r["area"] > sizes[k - 1] - r["steps"] / 2 - 1
    assert checked >= 30


@pytest.mark.parametrize("W,H", [(64, 48), (250, 130), (251, 77), (500, 300), (960, 540)])
@pytest.mark.parametrize("scale", [1.0, 4.0, -3.0])
@pytest.mark.parametrize("mode", ["box", "box_unstaged", "dense", "dense_staged", "dense_smallstage", "dense_unstaged", "dense_gather", "dense_boxes"])
def test_filter_mask_remap(torch_cuda, monkeypatch, W, H, scale, mode):
    """The remap variants -- the box kernel with its source pixels staged in LDS (the product path), the same with the
    taps taken from memory (what it does when a box's source region outgrows the LDS buffer), the dense kernel with its
    pipelined and its per-pixel gather (early-out off, tables beyond the compact format), and the box kernel on whole
    tiles -- against cv.undistort + filter as restated by the oracle; barrel, strong barrel, pincushion."""
    from gpu_util import unpack_mask
    torch = torch_cuda
    if mode == "box_unstaged":
        monkeypatch.setenv("MOCAP_BOX_STAGE_BYTES", "0")
    if mode.startswith("dense_") or mode == "dense":
        monkeypatch.setenv("MOCAP_GENERAL_FILTER", "1")  # the dense path: the row pipeline on the general 8-byte tables (pipelined gather)
    if mode in ("dense_staged", "dense_smallstage", "dense_unstaged"):
        monkeypatch.setenv("MOCAP_ROWS_STAGED", "1")      # ... on the compact table, source pixels through LDS
    if mode == "dense_smallstage":
        monkeypatch.setenv("MOCAP_ROWS_STAGE_DW", "600")  # ... with a buffer some bands' source rectangles outgrow
    if mode == "dense_unstaged":
        monkeypatch.setenv("MOCAP_ROWS_STAGE_DW", "0")    # ... every band's taps from memory
    if mode == "dense_gather":
        monkeypatch.setenv("MOCAP_REMAP_PIPELINE", "0")   # ... the general tables with the per-pixel gather
    if mode == "dense_boxes":
        monkeypatch.setenv("MOCAP_SKIP_DARK", "0")
        monkeypatch.setenv("MOCAP_DENSE_BOXES", "1")
    rng = np.random.default_rng(W + H)
    frames = rand_frames(rng, 2, H, W, bright=0.2, blobs=6)
    dist = np.array(MILD_DIST) * scale
    ctx, K, ident = make_ctx(W, H, dist=dist)
    assert not ident
    # the device-built tables reproduce the oracle's undistorted image, pixel for pixel
    und = ctx.undistort(torch.from_numpy(frames[0]).cuda()).cpu().numpy()
    assert np.array_equal(und, oracle.undistort(frames[0], K, dist))
    got, _ = unpack_mask(ctx.filter_mask(torch.from_numpy(frames).cuda()), W)
    for i in range(2):
        exp = oracle.image_filter(oracle.undistort(frames[i], K, dist), 0) != 0
        assert np.array_equal(got[i], exp)


def test_filter_mask_unaligned_pitch_and_threshold_variants(torch_cuda):
    from gpu_util import unpack_mask
    torch = torch_cuda
    rng = np.random.default_rng(5)
    H, W = 70, 123
    big = torch.from_numpy(rand_frames(rng, 2, H, W + 5)).cuda()
    view = big[:, :, 1:W + 1]  # pitch W+5, base offset 1: the byte-load path
    ctx, K, _ = make_ctx(W, H)
    for thresh in [255 * 0.85, 100.0, 0.0, 254.5, 255.0, 300.0, -5.0]:
        ctx.set_blob_params(thresh=thresh)
        got, _ = unpack_mask(ctx.filter_mask(view), W)
        for i in range(2):
            exp = oracle.image_filter(view[i].cpu().numpy(), 0, thresh=thresh) != 0
            assert np.array_equal(got[i], exp), thresh


def structured_mask(rng, H, W, n=10):
    yy, xx = np.mgrid[0:H, 0:W]
    m = np.zeros((H, W), bool)
    for _ in range(n):
        cx, cy, r = rng.uniform(0, W), rng.uniform(0, H), rng.uniform(4, min(H, W) / 3)
        d2 = (xx - cx) ** 2 + (yy - cy) ** 2
        m |= d2 <= r * r
        if rng.random() < 0.6:
            m &= ~(d2 <= (0.7 * r) ** 2)
            if rng.random() < 0.6:
                m |= d2 <= (0.4 * r) ** 2
                if rng.random() < 0.5:
                    m &= ~(d2 <= (0.2 * r) ** 2)
    return (m * 255).astype(np.uint8)


def check_against_oracle(mask, recs, xy, count, min_area, min_circ, max_blobs):
    """Every border the kernel found == every border cv.findContours would list (per the oracle), with the same
    measurements, the same parent, and the kept ones in the same order."""
    H, W = mask.shape
    table = oracle.find_contours(mask, min_area=min_area, min_circ=min_circ)
    for c in table:  # discovery position: the start pixel, or the background pixel right of it for a hole
        c["key"] = c["oy"] * (W + 1) + c["ox"] + (1 if c["is_hole"] else 0)
    by_key = {(r["key"], r["is_hole"]): r for r in recs}
    assert len(recs) == len(table) == len(by_key)
    for c in table:
        r = by_key[(c["key"], c["is_hole"])]
        for f in ("a00", "a10", "a01", "npts", "steps", "kept", "cx", "cy"):
            assert r[f] == c[f], (f, r, c)
        assert r["area"] == c["area"] and r["perimeter"] == c["perimeter"]
        assert (r["sx"], r["sy"]) == (c["ox"], c["oy"])
        assert (c["a00"] > 0) == bool(c["is_hole"]) or c["a00"] == 0  # orientation tells the border kind
        exp_parent = None if c["parent_order"] < 0 else (table[c["parent_order"]]["key"], table[c["parent_order"]]["is_hole"])
        got_parent = None if r["parent"] < 0 else (recs[r["parent"]]["key"], recs[r["parent"]]["is_hole"])
        assert exp_parent == got_parent, (c, r)
    kept = [c for c in table if c["kept"]]
    assert count == len(kept)
    for j, c in enumerate(kept):
        assert by_key[(c["key"], c["is_hole"])]["order"] == j
        if j < max_blobs:
            assert list(xy[j]) == [c["cx"], c["cy"]]
    return len(table), len(kept)


@pytest.mark.parametrize("seed", range(12))
def test_contours_match_oracle(torch_cuda, seed):
    from gpu_util import pack_mask
    from mocapv2_amd.engine import MocapContext
    rng = np.random.default_rng(seed)
    H, W = [(40, 70), (64, 64), (90, 130), (33, 31)][seed % 4]
    if seed % 3 == 0:
        masks = [((rng.random((H, W)) < p) * 255).astype(np.uint8) for p in (0.03, 0.9, 0.97)]
    else:
        masks = [structured_mask(rng, H, W, n=rng.integers(2, 9)) for _ in range(3)]
    masks.append(np.zeros((H, W), np.uint8))
    masks.append(np.full((H, W), 255, np.uint8))
    ctx = MocapContext(W, H)
    ctx.set_blob_params(min_area=20.0, min_circ=0.3)  # small gates so that many contours are kept
    xy, cnt, recs = ctx.contours_from_mask(pack_mask(np.stack(masks)), max_blobs=128, debug_cap=384)
    xy, cnt = xy.cpu().numpy(), cnt.cpu().numpy()
    seen = 0
    for i, m in enumerate(masks):
        if cnt[i] < 0:  # capacity overflow is reported, never silent
            assert len(oracle.find_contours(m)) > 100, cnt[i]
            continue
        n_all, n_kept = check_against_oracle(m, recs[i], xy[i], cnt[i], 20.0, 0.3, 128)
        seen += n_all
    assert seen > 0


@pytest.mark.parametrize("defer", [0, 2], ids=["links_in_place", "links_deferred"])
@pytest.mark.parametrize("seed", range(int(os.environ.get("MOCAP_FUZZ_CONTOUR_SEEDS", "6"))))
def test_contours_match_oracle_large_masks(torch_cuda, seed, defer):
    """The same comparison on masks several mask windows wide and tall (the follow kernel's lanes pause at the rim of their
    64 x 64 window and have it staged anew; the forward and the backward lane of a border leave their shared window at
    different times): rings within rings, long thin bars (borders walked out and back), blobs touching the image border,
    one batch of 12 images so that walks of different images share waves."""
    from gpu_util import pack_mask
    from mocapv2_amd.engine import MocapContext
    rng = np.random.default_rng(7000 + seed)
    H, W = [(300, 420), (257, 513), (480, 300)][seed % 3]
    masks = []
    for i in range(12):
        m = structured_mask(rng, H, W, n=rng.integers(3, 14)) != 0
        yy, xx = np.mgrid[0:H, 0:W]
        for _ in range(int(rng.integers(0, 4))):  # bars one or two pixels thick, any slope
            x0, y0, x1, y1 = rng.uniform(0, W), rng.uniform(0, H), rng.uniform(0, W), rng.uniform(0, H)
            t = ((xx - x0) * (x1 - x0) + (yy - y0) * (y1 - y0)) / max(1e-9, (x1 - x0) ** 2 + (y1 - y0) ** 2)
            d = np.hypot(xx - (x0 + t * (x1 - x0)), yy - (y0 + t * (y1 - y0)))
            m |= (d <= rng.choice([0.5, 0.8, 1.2])) & (t >= 0) & (t <= 1)
        if i % 4 == 3:
            m ^= rng.random((H, W)) < 0.002  # a few single-pixel specks and pinholes
        masks.append((m * 255).astype(np.uint8))
    ctx = MocapContext(W, H)
    ctx.set_blob_params(min_area=20.0, min_circ=0.05)
    ctx.set_tuning("contour_defer", defer)  # nested rings: links only a walk settles -- in the first tree pass, or through the second passes
    xy, cnt, recs = ctx.contours_from_mask(pack_mask(np.stack(masks)), max_blobs=128, debug_cap=384)
    xy, cnt = xy.cpu().numpy(), cnt.cpu().numpy()
    seen = longest = 0
    for i, m in enumerate(masks):
        if cnt[i] < 0:  # capacity overflow is reported, never silent
            assert len(oracle.find_contours(m)) > 100, cnt[i]
            continue
        n_all, _ = check_against_oracle(m, recs[i], xy[i], cnt[i], 20.0, 0.05, 128)
        seen += n_all
        longest = max([longest] + [r["steps"] for r in recs[i]])
    assert seen > 20 and longest > 300  # borders far longer than a window is wide were among them


@pytest.mark.parametrize("dist", [ZERO_DIST, MILD_DIST], ids=["nodist", "mild"])
def test_find_dot_synthetic_frames(torch_cuda, dist):
    torch = torch_cuda
    from mocapv2_amd.engine import MocapContext
    sc = Scene(3, width=960, height=540, dist=dist)
    frames = sc.render_batch(seed=42, n_steps=3, n_markers=6, radius_range=(16, 22), salt=0.001)  # [3,3,H,W]
    ctx = MocapContext(960, 540, n_slots=3)
    for s in range(3):
        ctx.set_undistort(s, sc.K, sc.dist)
    xy, cnt = ctx.record_views(ctx.blob_centroids(torch.from_numpy(frames).cuda(), cam_mod=3))
    xy, cnt = xy.cpu().numpy(), cnt.cpu().numpy()
    flat = frames.reshape(-1, 540, 960)
    total = 0
    for i in range(len(flat)):
        exp = oracle.find_dot(flat[i], sc.K, sc.dist)
        assert cnt[i] == len(exp)
        assert xy[i, :cnt[i]].tolist() == exp
        total += len(exp)
    assert total >= 40  # the scene really contains detectable markers


def dark_frames(rng, n, H, W, n_discs, salt, noise_max=60):
    """IR-like frames: dark noise, sparse salt pixels, a few saturated discs -- most tiles are provably all-zero,
    so the dark-tile early-out decides most of the mask."""
    img = rng.integers(0, noise_max + 1, (n, H, W), dtype=np.uint8)
    yy, xx = np.mgrid[0:H, 0:W]
    for i in range(n):
        k = int(salt * H * W)
        img[i, rng.integers(0, H, k), rng.integers(0, W, k)] = 255
        for _ in range(n_discs):
            cx, cy, r = rng.uniform(0, W), rng.uniform(0, H), rng.uniform(6, 24)
            d = np.sqrt((xx - cx) ** 2 + (yy - cy) ** 2)
            img[i] = np.maximum(img[i], (np.clip((r + 0.75 - d) / 1.5, 0, 1) * 255).astype(np.uint8))
    return img


@pytest.mark.parametrize("scale", [0.0, 1.0, -3.0])
@pytest.mark.parametrize("salt,noise_max", [(0.0, 60), (0.001, 60), (0.02, 60), (0.001, 140), (0.3, 60)])
def test_dark_tile_early_out_is_exact(torch_cuda, monkeypatch, scale, salt, noise_max):
    """With and without the early-out the mask equals the oracle's, for dark frames with sparse / dense salt noise,
    brighter backgrounds (the bound then rarely proves anything) and clusters of bright pixels right at the allowance."""
    from gpu_util import unpack_mask
    torch = torch_cuda
    W, H = 960, 540
    rng = np.random.default_rng(int(1000 * salt) + noise_max + int(10 * scale) + 7)
    frames = dark_frames(rng, 3, H, W, n_discs=5, salt=salt, noise_max=noise_max)
    # adversarial clusters: k saturated pixels packed into one cell block, k around the allowance (5..8), far from discs
    for i, k in enumerate([4, 6, 8, 12, 20]):
        y0, x0 = 40 + 90 * i, 30 + 7 * i
        pts = [(y0 + a, x0 + b) for a in range(5) for b in range(5)][:k]
        for (yy, xx) in pts:
            frames[0, yy, xx] = 255
    # plateaus barely above / below the threshold on a background at the darkest "bright-free" level: every pixel of the
    # plateau carries a small excess (217 - 63), none is near 255 -- the bound works on excess sums, not on counts
    frames[1, 200:260, 300:420] = 63
    frames[1, 215:227, 320:332] = 217   # box mean 217 > 216 inside: mask pixels
    frames[1, 215:227, 360:372] = 216   # never above the threshold
    frames[1, 236:241, 322:327] = 250   # too small for the median
    frames[1, :3, :] = 255      # saturated top rows (border taps counts)
    frames[2, :, W - 2:] = 255  # saturated right columns
    dist = np.array(MILD_DIST) * scale
    exp = None
    for skip in ["1", "0"]:
        monkeypatch.setenv("MOCAP_SKIP_DARK", skip)
        ctx, K, ident = make_ctx(W, H, dist=dist)
        got, pad = unpack_mask(ctx.filter_mask(torch.from_numpy(frames).cuda()), W)
        assert not pad.any()
        if exp is None:
            exp = [oracle.image_filter(frames[i] if ident else oracle.undistort(frames[i], K, dist), 0) != 0 for i in range(3)]
        for i in range(3):
            assert np.array_equal(got[i], exp[i]), (skip, i, np.argwhere(got[i] != exp[i])[:4])
    assert exp[0].any() and exp[1][200:260, 300:345].any() and not exp[1][200:260, 350:420].any()


@pytest.mark.parametrize("scale", [0.0, 1.0])
def test_context_mask_stays_consistent_across_batches(torch_cuda, scale):
    """The context's own mask is not cleared between batches: a tile filtered in one batch and dark in the next must
    be cleared on demand, a tile dark twice must stay untouched.  Batches with discs in different places, an all-dark
    batch, a shorter batch and a run with the early-out switched off are interleaved; every result must equal the
    oracle's _find_dot on the same frames."""
    torch = torch_cuda
    from mocapv2_amd.engine import MocapContext
    W, H = 960, 540
    sc = Scene(1, width=W, height=H, dist=np.array(MILD_DIST) * scale)
    ctx = MocapContext(W, H, n_slots=1)
    ctx.set_undistort(0, sc.K, sc.dist)
    rng = np.random.default_rng(77)
    batches = [dark_frames(rng, 4, H, W, n_discs=4, salt=0.001) for _ in range(3)]
    batches.insert(2, rng.integers(0, 50, (4, H, W), dtype=np.uint8))     # nothing bright at all
    batches.append(batches[0][:2])                                         # fewer images than before
    batches.append(dark_frames(rng, 4, H, W, n_discs=6, salt=0.0))
    seen = 0
    for b, frames in enumerate(batches):
        ctx.set_tuning("skip_dark", 0 if b == 4 else 1)  # one batch with the early-out off, on the live context
        xy, cnt = ctx.record_views(ctx.blob_centroids(torch.from_numpy(frames).cuda()))
        xy, cnt = xy.cpu().numpy(), cnt.cpu().numpy()
        for i in range(len(frames)):
            exp = oracle.find_dot(frames[i], sc.K, sc.dist)
            assert cnt[i] == len(exp), (b, i, cnt[i], exp)
            assert xy[i, :cnt[i]].tolist() == exp, (b, i)
            seen += len(exp)
    assert seen > 10


@pytest.mark.parametrize("scale", [0.0, 1.0])
def test_batches_of_varying_size_on_one_context(torch_cuda, scale):
    """Alternating n_images on one context (what BatchTracker.extract does per camera segment when world > 1): the boxes an
    earlier, larger batch left in the scan's alternate array for the images beyond a smaller batch must not survive into
    the next large one.  Results equal the oracle's, and the early-out resolves as many tiles in the last large batch as a
    fresh context does for the same frames (stale boxes would only cost tiles, never exactness)."""
    torch = torch_cuda
    from mocapv2_amd.engine import MocapContext
    W, H = 960, 540
    sc = Scene(1, width=W, height=H, dist=np.array(MILD_DIST) * scale)
    rng = np.random.default_rng(31)
    big = [dark_frames(rng, 6, H, W, n_discs=5, salt=0.0) for _ in range(4)]
    sizes = [6, 6, 2, 1, 6, 3, 6]
    ctx = MocapContext(W, H, n_slots=1)
    ctx.set_undistort(0, sc.K, sc.dist)
    stats = None
    for b, n in enumerate(sizes):
        frames = big[b % 4][:n]
        xy, cnt = ctx.record_views(ctx.blob_centroids(torch.from_numpy(frames).cuda()))
        xy, cnt = xy.cpu().numpy(), cnt.cpu().numpy()
        for i in range(n):
            exp = oracle.find_dot(frames[i], sc.K, sc.dist)
            assert cnt[i] == len(exp) and xy[i, :cnt[i]].tolist() == exp, (b, i)
        stats = ctx.tile_stats()
    fresh = MocapContext(W, H, n_slots=1)
    fresh.set_undistort(0, sc.K, sc.dist)
    fresh.blob_centroids(torch.from_numpy(big[(len(sizes) - 1) % 4][:sizes[-1]]).cuda())
    assert stats == fresh.tile_stats() and stats[1] > 0.5 * stats[0], (stats, fresh.tile_stats())


def test_a_lens_outside_the_sparse_paths_bounds_is_reported_not_silent(torch_cuda):
    """mocap_undistort_info: the identity and the mild lens take the sparse road; a lens whose 5x5 windows read more than
    9 x 9 source pixels falls back to the dense row pipeline -- same results (equal to the oracle), but the caller is told
    (sparse_path False, a RuntimeWarning from MocapContext.set_undistort) instead of silently running several times slower."""
    import warnings
    torch = torch_cuda
    from mocapv2_amd.engine import MocapContext
    W, H = 640, 360
    sc = Scene(1, width=W, height=H, dist=ZERO_DIST)
    ctx = MocapContext(W, H, n_slots=3)
    with warnings.catch_warnings():
        warnings.simplefilter("error")  # neither of these may warn
        ctx.set_undistort(0, sc.K, ZERO_DIST)
        ctx.set_undistort(1, sc.K, MILD_DIST)
    i0, i1 = ctx.undistort_info(0), ctx.undistort_info(1)
    assert i0["identity"] and i0["sparse_path"] and i0["max_source_weight"] == 1024
    assert not i1["identity"] and i1["sparse_path"] and i1["compact_table"] and i1["early_out_provable"] and i1["max_source_weight"] > 0
    wild = None
    for scale in (12.0, 30.0, 80.0):  # stronger and stronger barrel distortion until the bound is not provable any more
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            ctx.set_undistort(2, sc.K, np.array(MILD_DIST) * scale)
        if not ctx.undistort_info(2)["sparse_path"]:
            wild = np.array(MILD_DIST) * scale
            assert any(issubclass(w.category, RuntimeWarning) and "dense" in str(w.message) for w in caught)
            break
        assert not caught
    assert wild is not None, "no test lens left the sparse path"
    rng = np.random.default_rng(4)
    frames = dark_frames(rng, 2, H, W, n_discs=4, salt=0.001)
    xy, cnt = ctx.record_views(ctx.blob_centroids(torch.from_numpy(frames).cuda(), cam_mod=1, slot_base=2))
    xy, cnt = xy.cpu().numpy(), cnt.cpu().numpy()
    for i in range(2):
        exp = oracle.find_dot(frames[i], sc.K, wild)
        assert cnt[i] == len(exp) and xy[i, :cnt[i]].tolist() == exp


def test_set_tuning_rejects_unknown_names(torch_cuda):
    from mocapv2_amd._abi import MocapError
    from mocapv2_amd.engine import MocapContext
    ctx = MocapContext(64, 64)
    ctx.set_tuning("wide_bands", 3)
    with pytest.raises(MocapError):
        ctx.set_tuning("experiment_skip_box", 1)  # the work-skipping switches of round 2 are gone
    with pytest.raises(MocapError):
        ctx.set_tuning("rows", 32)  # shapes the context's buffers: environment only, before mocap_ctx_create


def test_mixed_identity_and_remapped_cameras_in_one_batch(torch_cuda):
    """cam_mod = 2 with one undistorted and one identity camera: both take the same launch (and the early-out)."""
    torch = torch_cuda
    from mocapv2_amd.engine import MocapContext
    W, H = 640, 360
    sc = Scene(2, width=W, height=H, dist=MILD_DIST)
    ctx = MocapContext(W, H, n_slots=2)
    ctx.set_undistort(0, sc.K, sc.dist)
    ctx.set_undistort(1, sc.K, ZERO_DIST)
    rng = np.random.default_rng(5)
    frames = dark_frames(rng, 6, H, W, n_discs=3, salt=0.001)
    xy, cnt = ctx.record_views(ctx.blob_centroids(torch.from_numpy(frames).cuda(), cam_mod=2))
    xy, cnt = xy.cpu().numpy(), cnt.cpu().numpy()
    total = 0
    for i in range(6):
        exp = oracle.find_dot(frames[i], sc.K, sc.dist if i % 2 == 0 else np.array(ZERO_DIST))
        assert cnt[i] == len(exp) and xy[i, :cnt[i]].tolist() == exp, i
        total += len(exp)
    assert total > 0
    tiles, skipped = ctx.tile_stats()
    assert 0 < skipped < tiles


@pytest.mark.parametrize("W,H", [(251, 131), (333, 77), (64, 48), (500, 300)])
@pytest.mark.parametrize("scale", [0.0, 2.0])
def test_centroids_odd_sizes_two_batches(torch_cuda, W, H, scale):
    """_find_dot through the context-owned mask (early-out, row ranges, on-demand clearing) at sizes that are not
    multiples of the 8x8 cells / 240x68 tiles, two different batches in a row."""
    torch = torch_cuda
    from mocapv2_amd.engine import MocapContext
    sc = Scene(1, width=W, height=H, dist=np.array(MILD_DIST) * scale)
    ctx = MocapContext(W, H, n_slots=1)
    ctx.set_undistort(0, sc.K, sc.dist)
    ctx.set_blob_params(min_area=60.0)
    prm = oracle.default_params(undistort=True)
    prm.min_area = 60.0
    rng = np.random.default_rng(W * 1000 + H)
    seen = 0
    for b in range(2):
        frames = dark_frames(rng, 3, H, W, n_discs=3, salt=0.002)
        xy, cnt = ctx.record_views(ctx.blob_centroids(torch.from_numpy(frames).cuda()))
        xy, cnt = xy.cpu().numpy(), cnt.cpu().numpy()
        for i in range(3):
            exp = oracle.find_dot(frames[i], sc.K, sc.dist, params=prm)
            assert cnt[i] == len(exp) and xy[i, :cnt[i]].tolist() == exp, (b, i, cnt[i], exp)
            seen += len(exp)
    assert seen > 0


@pytest.mark.parametrize("seed", range(int(os.environ.get("MOCAP_FUZZ_SEEDS", "12"))))
def test_early_out_fuzz_thresholds_and_brightness(torch_cuda, seed):
    """Random thresholds, background levels, bright clutter and lens models: the mask (caller-owned, scan-cleared) and
    the centroids (context-owned mask, cleared on demand) equal the oracle's, batch after batch on one context."""
    from gpu_util import unpack_mask
    torch = torch_cuda
    from mocapv2_amd.engine import MocapContext
    rng = np.random.default_rng(9000 + seed)
    W, H = int(rng.choice([256, 488, 640, 964])), int(rng.choice([96, 200, 273, 360]))
    scale = float(rng.choice([0.0, 0.5, 1.0, 3.0, -2.0]))
    dist = np.array(MILD_DIST) * scale
    sc = Scene(1, width=W, height=H, dist=dist)
    ctx = MocapContext(W, H, n_slots=1)
    ident = ctx.set_undistort(0, sc.K, sc.dist)
    for b in range(3):
        thresh = float(rng.choice([255 * 0.85, 130.0, 180.0, 240.0, 90.0]))
        noise_max = int(rng.choice([20, 60, 63, 64, 100]))
        frames = dark_frames(rng, 2, H, W, n_discs=int(rng.integers(0, 5)), salt=float(rng.choice([0.0, 0.001, 0.01])),
                             noise_max=noise_max)
        # clutter: small saturated squares of 2..6 px, some at the image border
        for i in range(2):
            for _ in range(int(rng.integers(0, 12))):
                k = int(rng.integers(2, 7))
                y0 = int(rng.integers(0, H - k)) if rng.random() < 0.7 else int(rng.choice([0, H - k]))
                x0 = int(rng.integers(0, W - k)) if rng.random() < 0.7 else int(rng.choice([0, W - k]))
                frames[i, y0:y0 + k, x0:x0 + k] = 255
        ctx.set_blob_params(thresh=thresh, min_area=40.0)
        prm = oracle.default_params(undistort=True)
        prm.thresh = thresh
        prm.min_area = 40.0
        d = torch.from_numpy(frames).cuda()
        got, pad = unpack_mask(ctx.filter_mask(d), W)
        assert not pad.any()
        xy, cnt = ctx.record_views(ctx.blob_centroids(d))
        xy, cnt = xy.cpu().numpy(), cnt.cpu().numpy()
        for i in range(2):
            und = frames[i] if ident else oracle.undistort(frames[i], sc.K, dist)
            exp = oracle.image_filter(und, 0, thresh=thresh) != 0
            assert np.array_equal(got[i], exp), (seed, b, i, thresh, noise_max, np.argwhere(got[i] != exp)[:4])
            pts = oracle.find_dot(frames[i], sc.K, dist, params=prm)
            assert cnt[i] == len(pts) and xy[i, :cnt[i]].tolist() == pts, (seed, b, i)


@pytest.mark.parametrize("seed", range(int(os.environ.get("MOCAP_FUZZ_LARGE_SEEDS", "6"))))
def test_box_items_fuzz_large_frames(torch_cuda, seed):
    """Full-size frames aimed at the work list of the sparse filter: discs of 3..70 px radius (boxes split into parts,
    boxes too wide for an item -> the row pipeline), rings, discs centred on tile corners (x = 240 k, y = 68 k: the 2x2
    tile clusters) and on the image border, three cameras with different lens tables in one time-major batch, two
    batches per context (the on-demand clearing of the previous batch's boxes).  Mask and centroids = the oracle's."""
    from gpu_util import unpack_mask
    torch = torch_cuda
    from mocapv2_amd.engine import MocapContext
    rng = np.random.default_rng(31000 + seed)
    W, H = [(1920, 1080), (1280, 720), (1924, 1082)][seed % 3]
    C = 3
    scales = [float(rng.choice([0.0, 1.0, 2.5, -2.0])) for _ in range(C)]
    sc = Scene(1, width=W, height=H)
    ctx = MocapContext(W, H, n_slots=C)
    ident = [ctx.set_undistort(c, sc.K, np.array(MILD_DIST) * scales[c]) for c in range(C)]
    yy, xx = np.mgrid[0:H, 0:W]
    seen = 0
    for b in range(2):
        frames = rng.integers(0, 61, (2 * C, H, W), dtype=np.uint8)
        for i in range(2 * C):
            for _ in range(int(rng.integers(1, 9))):
                kind = rng.random()
                r = float(rng.uniform(3, 24)) if kind < 0.6 else float(rng.uniform(24, 70))
                where = rng.random()
                if where < 0.35:   # on a tile corner (+- a few pixels)
                    cx = 240.0 * int(rng.integers(1, max(2, W // 240))) + rng.uniform(-6, 6)
                    cy = 68.0 * int(rng.integers(1, max(2, H // 68))) + rng.uniform(-6, 6)
                elif where < 0.5:  # on the image border
                    cx, cy = (float(rng.choice([0, W - 1])), rng.uniform(0, H)) if rng.random() < 0.5 else (rng.uniform(0, W), float(rng.choice([0, H - 1])))
                else:
                    cx, cy = rng.uniform(0, W), rng.uniform(0, H)
                x0, x1 = max(0, int(cx - r - 3)), min(W, int(cx + r + 4))
                y0, y1 = max(0, int(cy - r - 3)), min(H, int(cy + r + 4))
                if x0 >= x1 or y0 >= y1:
                    continue
                d = np.sqrt((xx[y0:y1, x0:x1] - cx) ** 2 + (yy[y0:y1, x0:x1] - cy) ** 2)
                disc = (np.clip((r + 0.75 - d) / 1.5, 0, 1) * 255).astype(np.uint8)
                if rng.random() < 0.25:  # a ring: hole borders
                    disc[d < 0.5 * r] = 0
                frames[i, y0:y1, x0:x1] = np.maximum(frames[i, y0:y1, x0:x1], disc)
            k = int(rng.choice([0, 200, 2000]))
            frames[i, rng.integers(0, H, k), rng.integers(0, W, k)] = 255
        d = torch.from_numpy(frames).cuda()
        got, pad = unpack_mask(ctx.filter_mask(d, cam_mod=C), W)
        assert not pad.any()
        xy, cnt = ctx.record_views(ctx.blob_centroids(d, cam_mod=C))
        xy, cnt = xy.cpu().numpy(), cnt.cpu().numpy()
        for i in range(2 * C):
            c = i % C
            dist = np.array(MILD_DIST) * scales[c]
            und = frames[i] if ident[c] else oracle.undistort(frames[i], sc.K, dist)
            exp = oracle.image_filter(und, 0) != 0
            assert np.array_equal(got[i], exp), (seed, b, i, scales[c], np.argwhere(got[i] != exp)[:4])
            pts = oracle.find_dot(frames[i], sc.K, dist)
            assert cnt[i] == len(pts) and xy[i, :cnt[i]].tolist() == pts, (seed, b, i)
            seen += len(pts)
    assert seen > 0


@pytest.mark.parametrize("W,H", [(16000, 48), (40, 5000), (4096, 72)])
def test_extreme_aspect_ratios(torch_cuda, W, H):
    """More than 64 strips per chunk (lane-parallel tile test in two rounds), many chunk groups, patches per tile."""
    torch = torch_cuda
    from mocapv2_amd.engine import MocapContext
    sc = Scene(1, width=W, height=H, dist=np.array(MILD_DIST) * 0.2)
    ctx = MocapContext(W, H, n_slots=1)
    ctx.set_undistort(0, sc.K, sc.dist)
    ctx.set_blob_params(min_area=60.0)
    prm = oracle.default_params(undistort=True)
    prm.min_area = 60.0
    rng = np.random.default_rng(W + H)
    seen = 0
    for b in range(2):
        frames = rng.integers(0, 60, (2, H, W), dtype=np.uint8)
        yy, xx = np.mgrid[0:H, 0:W]
        for i in range(2):
            for _ in range(6):
                cx, cy, r = rng.uniform(0, W), rng.uniform(0, H), rng.uniform(6, 14)
                x0, x1 = max(0, int(cx - r - 2)), min(W, int(cx + r + 3))
                y0, y1 = max(0, int(cy - r - 2)), min(H, int(cy + r + 3))
                d = np.sqrt((xx[y0:y1, x0:x1] - cx) ** 2 + (yy[y0:y1, x0:x1] - cy) ** 2)
                frames[i, y0:y1, x0:x1] = np.maximum(frames[i, y0:y1, x0:x1], (np.clip((r + 0.75 - d) / 1.5, 0, 1) * 255).astype(np.uint8))
        xy, cnt = ctx.record_views(ctx.blob_centroids(torch.from_numpy(frames).cuda()))
        xy, cnt = xy.cpu().numpy(), cnt.cpu().numpy()
        for i in range(2):
            exp = oracle.find_dot(frames[i], sc.K, sc.dist, params=prm)
            assert cnt[i] == len(exp) and xy[i, :cnt[i]].tolist() == exp, (b, i, cnt[i], len(exp))
            seen += len(exp)
    assert seen > 0


# ---- Bayer -> gray in front of the path (RealtimeTracking_FLIR.py:103-104) --------------------------------------------
@pytest.mark.parametrize("pattern,shift", [(3, 14), (3, 15), (0, 14), (1, 14), (2, 15)])
def test_bayer_gray_matches_oracle(pattern, shift):
    import torch
    from mocapv2_amd.engine import MocapContext
    ctx = MocapContext(8, 8)
    rng = np.random.default_rng(60 + pattern)
    # 16- and 8-pixel-per-lane kernels: widths that are multiples of 16 / of 8, one or several waves per row, partial last
    # wave; generic kernel: the rest
    for H, W in ((3, 16), (8, 256), (30, 1920), (9, 2000), (5, 992), (3, 8), (12, 1000), (7, 496), (6, 504), (3, 4), (21, 260),
                 (3, 3), (5, 7), (37, 29), (16, 258)):
        raw = rng.integers(0, 256, (H, W), dtype=np.uint8)
        got = ctx.bayer_gray(torch.from_numpy(raw).cuda(), pattern, shift).cpu().numpy()
        assert np.array_equal(got, oracle.bayer_gray(raw, pattern, shift)), (H, W)
    # a batch, rows and images of the source with padding (pitch 272, image stride 20 rows)
    n, H, W = 3, 18, 264
    buf = rng.integers(0, 256, (n, 20, 272), dtype=np.uint8)
    d = torch.from_numpy(buf).cuda()
    got = ctx.bayer_gray(d[:, :H, :W], pattern, shift).cpu().numpy()
    for i in range(n):
        assert np.array_equal(got[i], oracle.bayer_gray(buf[i, :H, :W], pattern, shift))
    # an unaligned view takes the generic kernel
    got = ctx.bayer_gray(d[:, 1:1 + H, 1:1 + W], pattern, shift).cpu().numpy()
    for i in range(n):
        assert np.array_equal(got[i], oracle.bayer_gray(buf[i, 1:1 + H, 1:1 + W], pattern, shift))


def test_bayer_gray_rejects_bad_arguments():
    import torch
    from mocapv2_amd._abi import MocapError
    from mocapv2_amd.engine import MocapContext
    ctx = MocapContext(8, 8)
    raw = torch.zeros((8, 8), dtype=torch.uint8, device="cuda")
    for kw in ({"pattern": 4}, {"pattern": -1}, {"gray_shift": 16}):
        with pytest.raises(MocapError):
            ctx.bayer_gray(raw, **kw)
    with pytest.raises(MocapError):
        ctx.bayer_gray(raw[:2], 3)


def test_replay_from_raw_bayer_frames():
    """Raw sensor frames through the tracker = the same tracker fed with the oracle's gray conversion of them."""
    from mocapv2_amd.pipeline import scene_arrays
    from mocapv2_amd.replay import ReplayTracker
    sc = Scene(3, width=640, height=360, dist=MILD_DIST)
    T = 5
    gray_scene = sc.render_batch(77, T, 4, radius_range=(16, 19), salt=0.001)   # [T, C, H, W]
    rng = np.random.default_rng(3)
    # a sensor image whose demosaiced luma shows the same bright discs: scale the colour sites a little differently
    raw = gray_scene.astype(np.float64)
    raw[:, :, 0::2, 1::2] *= 0.9
    raw[:, :, 1::2, 0::2] *= 0.95
    raw = np.clip(raw + rng.integers(0, 3, raw.shape), 0, 255).astype(np.uint8)
    from mocapv2_amd.engine import GRAY_SHIFT
    gray = np.stack([np.stack([oracle.bayer_gray(raw[t, c], 3, GRAY_SHIFT) for c in range(3)]) for t in range(T)])
    arrays = scene_arrays(sc)
    a = list(ReplayTracker(*arrays, 640, 360, batch=4, bayer_pattern=3).run(raw))
    b = list(ReplayTracker(*arrays, 640, 360, batch=4).run(gray))
    assert len(a) == len(b) == T
    for x, y in zip(a, b):
        assert np.array_equal(x["object_points"], y["object_points"]) and np.array_equal(x["image_points"], y["image_points"])
        assert x["message"] == y["message"]
    assert sum(len(x["object_points"]) > 0 for x in a) >= T - 1


@pytest.mark.parametrize("W,H", [(640, 360), (1920, 1080), (648, 364), (96, 64)])
@pytest.mark.parametrize("scale", [0.0, 1.0])
def test_blob_centroids_from_bayer_equals_the_two_steps(torch_cuda, monkeypatch, W, H, scale):
    """mocap_blob_centroids_bayer (gray conversion fused with the early-out's scan where width % 16 == 0 and
    height % 8 == 0, separate kernels otherwise) = mocap_bayer_gray_u8 followed by mocap_blob_centroids, and the gray
    frames it leaves behind are the oracle's; two batches on one context, then once more with the early-out off."""
    from mocapv2_amd.engine import GRAY_SHIFT
    torch = torch_cuda
    from mocapv2_amd.engine import MocapContext
    rng = np.random.default_rng(W + H + int(scale))
    dist = np.array(MILD_DIST) * scale
    sc = Scene(2, width=W, height=H, dist=dist)
    n_discs = 4 if W >= 640 else 1
    for skip in ("1", "0"):
        monkeypatch.setenv("MOCAP_SKIP_DARK", skip)
        ctx = MocapContext(W, H, n_slots=2)
        ref = MocapContext(W, H, n_slots=2)
        for sl in range(2):
            ctx.set_undistort(sl, sc.K, sc.dist)
            ref.set_undistort(sl, sc.K, sc.dist)
        ctx.set_blob_params(min_area=60.0)
        ref.set_blob_params(min_area=60.0)
        for b in range(2):
            raw = dark_frames(rng, 4, H, W, n_discs=n_discs, salt=0.002, noise_max=60)
            raw[:, 0::2, 1::2] = (raw[:, 0::2, 1::2] * 0.9).astype(np.uint8)   # the colour sites respond differently
            d = torch.from_numpy(raw).cuda()
            gray = torch.zeros_like(d)
            rec = ctx.blob_centroids(d, cam_mod=2, bayer_pattern=3, gray=gray).cpu().numpy()
            two = ref.blob_centroids(ref.bayer_gray(d, 3), cam_mod=2).cpu().numpy()
            g = gray.cpu().numpy()
            for i in range(4):
                assert np.array_equal(g[i], oracle.bayer_gray(raw[i], 3, GRAY_SHIFT)), (skip, b, i)
                n = rec[i, 0]
                assert n == two[i, 0] and n >= 0 and np.array_equal(rec[i, 2:2 + 2 * n], two[i, 2:2 + 2 * n]), (skip, b, i)
            assert rec[:, 0].sum() >= (4 if W >= 640 else 0)


@pytest.mark.parametrize("env", [
    {"MOCAP_WIDE_QUADS": "0,0"},                              # every marked tile through the row pipeline's list form
    {"MOCAP_WIDE_QUADS": "1000,1000"},                        # every marked tile through the box kernel
    {"MOCAP_WIDE_QUADS": "0,0", "MOCAP_WIDE_BANDS": "3"},     # ... its rows cut into bands
    {"MOCAP_WIDE_QUADS": "0,0", "MOCAP_ROWS_STAGED": "1"},    # ... on the compact table, source pixels staged in LDS
    {"MOCAP_WIDE_QUADS": "0,0", "MOCAP_ROWS_STAGED": "1", "MOCAP_ROWS_STAGE_DW": "500"},  # ... with a buffer too small for some bands
    {"MOCAP_WIDE_QUADS": "0,0", "MOCAP_WIDE_BLOCKS_PER_CU": "4"},
    {"MOCAP_WIDE_QUADS": "0,0", "MOCAP_WIDE_FORK": "1"},      # ... on the side stream beside the box kernel
    {"MOCAP_EXCESS_BASE": "0"}, {"MOCAP_EXCESS_BASE": "100"}, {"MOCAP_EXCESS_BASE": "200"},  # pinned excess bases
    {"MOCAP_BASE_SEL": "0"},                                  # start with the tight base (then adapt)
    {"MOCAP_BOX_BLOCKS_PER_CU": "1"},
    {"MOCAP_CONTOURS_SPLIT": "0"},                            # the contour stage as one kernel per image
    {"MOCAP_CONTOUR_BOXES": "0"},                             # candidates from whole strips instead of the tiles' boxes
    {"MOCAP_CONTOURS_SPLIT": "0", "MOCAP_CONTOUR_BOXES": "0"},
    {"MOCAP_SCAN_BLOCKS_PER_CU": "3"},                        # the scan as a persistent pass (blocks from a shared counter)
    {"MOCAP_SCAN_SLICES": "5"},                               # the scan in five launches over runs of images
    {"MOCAP_SCAN_SLICES": "3", "MOCAP_SCAN_BLOCKS_PER_CU": "2", "MOCAP_SCAN_WIDE": "0"},
    {"MOCAP_CORR_THREADS": "64"},
    {"MOCAP_SCAN_SERIAL": "0"},                               # scans of different contexts may share the chip
    {"MOCAP_SCAN_HOTMAP": "0"},                               # the scan marks the tiles itself instead of leaving the hot map
    {"MOCAP_SCAN_HOTMAP": "0", "MOCAP_SCAN_WIDE": "0"},
    {"MOCAP_SCAN_HOTMAP": "2"},                               # the scan always leaves the hot map (default: only for crowded scenes)
    {"MOCAP_SCAN_HOTMAP": "2", "MOCAP_SCAN_WIDE": "0"},       # ... from the 8-byte-load form of the scan (16 lanes per word)
    {"MOCAP_SCAN_HOTMAP": "2", "MOCAP_MARK_BLOCKS_PER_CU": "1"},  # ... read by a fixed grid of looping waves
    {"MOCAP_CONTOUR_BLOCKS_PER_CU": "1"},                     # the per-image contour kernels as a fixed grid looping over the images
    {"MOCAP_CONTOUR_DEFER": "2"},                             # links that need a walk always through the second follow / tree passes
    {"MOCAP_CONTOUR_DEFER": "0"},                             # ... always walked in place by the first tree pass
], ids=lambda e: ",".join(f"{k[6:]}={v}" for k, v in e.items()))
def test_tuning_switches_do_not_change_results(torch_cuda, monkeypatch, env):
    """Routing thresholds, row bands, the side stream, the scan's excess base (pinned or adapting), the grid size and the
    form of the contour stage are performance knobs: centroids and masks equal the oracle's whatever they are.  Three batches on one context -- dark
    frames, frames with a bright background (where the tolerant base is the better one), dark frames again -- so that the
    base adapts in between and the mask's "zero outside the recorded regions" invariant is exercised across the switch."""
    torch = torch_cuda
    from gpu_util import unpack_mask
    from mocapv2_amd.engine import MocapContext
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    W, H = 640, 360
    sc = Scene(2, width=W, height=H, dist=np.array(MILD_DIST))
    ctx = MocapContext(W, H, n_slots=2)
    for sl in range(2):
        ctx.set_undistort(sl, sc.K, sc.dist)
    rng = np.random.default_rng(17)
    for b, (lo, hi) in enumerate([(0, 60), (95, 115), (0, 60), (0, 60)]):
        frames = np.stack([sc.render(rng, sc.markers(rng, 5, extent=0.8), i % 2, radius_range=(15, 20), noise_min=lo, noise_max=hi, salt=0.001)
                           for i in range(34)])  # 34 images: every 16th one is probed
        dev = torch.from_numpy(frames).cuda()
        rec = ctx.blob_centroids(dev, cam_mod=2).cpu().numpy()
        got, _ = unpack_mask(ctx.filter_mask(dev, cam_mod=2), W)
        for i in range(len(frames)):
            exp, m = oracle.find_dot(frames[i], sc.K, sc.dist, return_mask=True)
            n = rec[i, 0]
            assert n == len(exp) and rec[i, 2:2 + 2 * n].reshape(-1, 2).tolist() == exp, (b, i, env)
            assert np.array_equal(got[i], m != 0), (b, i, env)


def test_translation_invariance_identity_lens(torch_cuda):
    """A size- and oracle-independent property of the whole blob stage: with the identity lens map, moving a frame's content by (dx, dy)
    moves the mask and every centroid by exactly (dx, dy) and keeps the contour order -- whatever 8 x 8 scan cell, 240 x 68 filter tile,
    mask byte / word, box-kernel item or wide-tile entry the blobs then fall into.  One 1080p frame (discs, a merged pair, a ring, a
    bar; noise background) in 14 shifted copies as one batch."""
    torch = torch_cuda
    from gpu_util import unpack_mask
    from mocapv2_amd.engine import MocapContext
    H, W, M = 1080, 1920, 320  # M: margin the content keeps from the border, so that no shift cuts a blob
    rng = np.random.default_rng(77)
    base = rng.integers(0, 50, (H, W), dtype=np.uint8)
    yy, xx = np.mgrid[0:H, 0:W]
    centres = []
    while len(centres) < 14:
        cx, cy, r = rng.integers(M, W - M), rng.integers(M, H - M), rng.integers(14, 26)
        if all((cx - a) ** 2 + (cy - b) ** 2 > (r + c + 12) ** 2 for a, b, c in centres):
            centres.append((cx, cy, r))
            base[(xx - cx) ** 2 + (yy - cy) ** 2 <= r * r] = 255
    cx, cy, r = centres[0]
    base[(xx - cx - r - 8) ** 2 + (yy - cy) ** 2 <= r * r] = 255                                  # a merged pair
    cx, cy, r = centres[1]
    base[((xx - cx) ** 2 + (yy - cy) ** 2 <= (r + 14) ** 2) & ((xx - cx) ** 2 + (yy - cy) ** 2 >= (r + 4) ** 2)] = 255  # a ring around a disc
    base[M - 40:M - 28, M:M + 300] = 255                                                           # a bar: a wide box
    shifts = [(0, 0), (1, 0), (0, 1), (3, 5), (7, 7), (8, 8), (31, 33), (32, 64), (239, 67), (240, 68), (-1, -1), (-9, -70), (-241, 5), (121, -35)]
    batch = np.zeros((len(shifts), H, W), np.uint8)
    for i, (dx, dy) in enumerate(shifts):
        ys, yd = (slice(0, H - dy), slice(dy, H)) if dy >= 0 else (slice(-dy, H), slice(0, H + dy))
        xs, xd = (slice(0, W - dx), slice(dx, W)) if dx >= 0 else (slice(-dx, W), slice(0, W + dx))
        batch[i][yd, xd] = base[ys, xs]
    ctx = MocapContext(W, H, 1)
    ctx.set_undistort(0, np.array([[1000.0, 0, W / 2], [0, 1000.0, H / 2], [0, 0, 1]]), np.zeros(5))
    dev = torch.from_numpy(batch).cuda()
    masks, _ = unpack_mask(ctx.filter_mask(dev, cam_mod=1), W)
    xy, cnt = ctx.record_views(ctx.blob_centroids(dev, cam_mod=1))
    xy, cnt = xy.cpu().numpy(), cnt.cpu().numpy()
    assert cnt[0] >= 14 and masks[0].sum() > 10000
    inner = (slice(250, H - 250), slice(250, W - 250))  # the part of the mask every shifted copy still holds
    for i, (dx, dy) in enumerate(shifts):
        assert cnt[i] == cnt[0], (dx, dy)
        assert np.array_equal(xy[i, :cnt[i]], xy[0, :cnt[0]] + np.array([dx, dy])), (dx, dy)
        moved = np.zeros_like(masks[0])
        ys, yd = (slice(0, H - dy), slice(dy, H)) if dy >= 0 else (slice(-dy, H), slice(0, H + dy))
        xs, xd = (slice(0, W - dx), slice(dx, W)) if dx >= 0 else (slice(-dx, W), slice(0, W + dx))
        moved[yd, xd] = masks[0][ys, xs]
        assert np.array_equal(masks[i][inner], moved[inner]), (dx, dy)


def test_crowded_scenes_switch_the_marking_to_the_hot_map_and_back(torch_cuda):
    """scan_hotmap = 1 (the default): who turns hot cells into tile boxes follows the scene -- the scan itself on sparse frames, the hot
    map + mark_tiles_kernel once a probe has counted more than ~400 hot cells per image.  One context sees sparse batches, crowded
    ones and sparse ones again; the probe runs on the first batch and on every 32nd after it and flips the mode in between.  Masks
    and centroids equal the oracle's on the batches around the switches, whatever marked the tiles: nothing of one mode's state (hot
    map words, tile boxes of either alternating array) leaks into the other's batches."""
    torch = torch_cuda
    from gpu_util import unpack_mask
    from mocapv2_amd.engine import MocapContext
    W, H = 640, 360
    sc = Scene(2, width=W, height=H, dist=np.array(MILD_DIST))
    ctx = MocapContext(W, H, n_slots=2)
    for sl in range(2):
        ctx.set_undistort(sl, sc.K, sc.dist)
    rng = np.random.default_rng(23)
    sparse = [dark_frames(rng, 34, H, W, n_discs=3, salt=0.001) for _ in range(3)]
    crowded = [dark_frames(rng, 34, H, W, n_discs=30, salt=0.001) for _ in range(3)]
    # 74 batches: probes on batch 0 (sparse), 32 (crowded: the map takes over) and 64 (sparse again: back to the scan)
    plan = [sparse] * 4 + [crowded] * 56 + [sparse] * 14
    checked = set(range(0, 4)) | {32, 33, 34, 35, 59, 60, 61, 64, 65, 66, 67, 73}
    for b, pool in enumerate(plan):
        frames = pool[b % 3]
        dev = torch.from_numpy(frames).cuda()
        rec = ctx.blob_centroids(dev, cam_mod=2).cpu().numpy()
        if b in checked:
            got, _ = unpack_mask(ctx.filter_mask(dev, cam_mod=2), W)
            for i in range(b % 5, len(frames), 5):
                exp, m = oracle.find_dot(frames[i], sc.K, sc.dist, return_mask=True)
                n = rec[i, 0]
                assert n == len(exp) and rec[i, 2:2 + 2 * n].reshape(-1, 2).tolist() == exp, (b, i)
                assert np.array_equal(got[i], m != 0), (b, i)


@pytest.mark.parametrize("W,H", [(64, 48), (250, 130), (251, 77), (500, 300), (961, 541), (1920, 1080)])
@pytest.mark.parametrize("scale", [0.0, 1.0])
def test_hot_map_marking_at_any_size(torch_cuda, W, H, scale):
    """The hot map (two bits per 8x8 source cell, written by whole waves of the scan, read 256 words per wave by mark_tiles_kernel) on
    frame sizes that end in part-cells, part-words and part-waves, in both forms of the scan (16-byte loads where the width allows,
    8-byte loads otherwise), with and without a lens table: masks and centroids equal the oracle's, and equal what the scan's own
    marking gives on the same context right afterwards."""
    from gpu_util import unpack_mask
    torch = torch_cuda
    from mocapv2_amd.engine import MocapContext
    rng = np.random.default_rng(W * 3 + H)
    sc = Scene(1, width=W, height=H, dist=np.array(MILD_DIST) * scale)
    ctx = MocapContext(W, H, n_slots=1)
    ctx.set_undistort(0, sc.K, sc.dist)
    n = 5 if W < 1000 else 2
    frames = dark_frames(rng, n, H, W, n_discs=max(2, W * H // 40000), salt=0.001)
    dev = torch.from_numpy(frames).cuda()
    results = {}
    for mode in (2, 0, 2):
        ctx.set_tuning("scan_hotmap", mode)
        rec = ctx.blob_centroids(dev).cpu().numpy()
        got, _ = unpack_mask(ctx.filter_mask(dev), W)
        results.setdefault(mode, []).append((rec.copy(), got.copy()))
    for i in range(n):
        exp, m = oracle.find_dot(frames[i], sc.K, sc.dist, return_mask=True)
        for mode, runs in results.items():
            for rec, got in runs:
                k = rec[i, 0]
                assert k == len(exp) and rec[i, 2:2 + 2 * k].reshape(-1, 2).tolist() == exp, (mode, i)
                assert np.array_equal(got[i], m != 0), (mode, i)
