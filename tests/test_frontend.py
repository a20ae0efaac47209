"""Patch front-end and the config-1 plumbing on the reference's own frames
(tests/golden/frames/*.ppm are data files of the reference's datasets/test)."""
import glob
import os
from collections import namedtuple

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import patches as opatch

FRAMES = sorted(glob.glob(os.path.join(GOLDEN, "frames", "*.ppm")))
KP = namedtuple("KP", "pt response")


def test_oracle_window_rule():
    """get_1d_boundaries (CvInputParser.py:49-89): shift forward below 0, back above dim-1."""
    c = np.array([[0, 0], [20, 20], [5, 235], [191, 239], [100, 120]])
    lo, hi = opatch.get_1d_boundaries((192, 240), c, 41, 0)
    assert lo.tolist() == [0, 0, 0, 151, 80] and hi.tolist() == [40, 40, 40, 191, 120]
    lo, hi = opatch.get_1d_boundaries((192, 240), c, 41, 1)
    assert lo.tolist() == [0, 0, 199, 199, 100] and hi.tolist() == [40, 40, 239, 239, 140]
    with pytest.raises(ValueError):
        opatch.get_1d_boundaries((192, 240), c, 40, 0)
    with pytest.raises(ValueError):
        opatch.get_1d_boundaries((192, 240), np.zeros((3, 3)), 41, 0)


def test_oracle_reads_reference_frames():
    assert len(FRAMES) == 3
    img = opatch.read_ppm(FRAMES[0])
    assert img.shape == (192, 240, 3) and img.dtype == np.uint8
    g = opatch.bgr2gray_opencv(img)
    assert g.shape == (192, 240) and 0 < g.mean() < 255
    p = opatch.parse(g, [(96.5, 120.5), (0, 0)], 41)
    assert p.shape == (2, 1681) and np.array_equal(p[1], (g[0:41, 0:41].reshape(-1) / 255.0))
    assert np.array_equal(p[0], g[76:117, 100:141].reshape(-1) / 255.0)      # round(96.5)=96, round(120.5)=120


@pytest.mark.gpu
def test_patch_frontend_vs_oracle():
    import deeploopcloser_amd as dlc
    parser = dlc.CvInputParser()
    rng = np.random.RandomState(0)
    for path in FRAMES:
        rgb = opatch.read_ppm(path)
        gray = opatch.bgr2gray_opencv(rgb)
        assert np.array_equal(dlc.default_engine().rgb_to_gray(torch.from_numpy(rgb.copy()).cuda()).cpu().numpy(), gray)
        pts = dlc.grid_key_points(gray.shape, 24) + [(-3.2, 500.0), (191.5, 239.5), (0.5, 1.5), (2.5, 3.5), (95.49, 10), (10, 229.51)]
        got = parser.parse(gray, pts)
        assert got.dtype == np.float64 and got.shape == (30, 1681)
        assert np.array_equal(got, opatch.parse(gray, pts, 41))
        assert np.array_equal(parser.parse_from_path(path, pts), got)             # RGB file -> grey on the GPU
        # cv2.KeyPoint-like objects: descending response, top n (CvInputParser.py:45-46)
        kps = [KP(pt=(float(rng.uniform(0, 191)), float(rng.uniform(0, 239))), response=float(rng.rand())) for _ in range(50)]
        want = opatch.parse(gray, [k.pt for k in sorted(kps, key=lambda k: -k.response)[:30]], 41)
        assert np.array_equal(parser.parse(gray, kps), want)
    assert parser.parse(gray, []).shape == (0, 1681)
    with pytest.raises(ValueError):
        dlc.CvInputParser(patch_size=40).parse(gray, pts)


@pytest.mark.gpu
def test_config1_plumbing_frames_to_matrices():
    """configs[0]: reference frames -> patches -> SDAV descriptors -> cosine matrix + SDAV
    similarity matrix, GPU vs oracle end to end."""
    import deeploopcloser_amd as dlc
    from oracle import sdav as osdav, similarity as osim, cosine as ocos
    parser = dlc.CvInputParser()
    net = dlc.SDAV(seed=4)
    ws, bs = net.get_weights()
    x = np.stack([parser.parse_from_path(p, dlc.grid_key_points((192, 240), 30)) for p in FRAMES])
    xo = np.stack([opatch.parse(opatch.bgr2gray_opencv(opatch.read_ppm(p)), dlc.grid_key_points((192, 240), 30)) for p in FRAMES])
    assert np.array_equal(x, xo) and x.shape == (3, 30, 1681)
    h = dlc.encode(x, net)
    ho = osdav.transform(xo, ws, bs)
    assert h.shape == (90, 2500) and np.abs(h - ho).max() < 1e-10
    frame_desc = dlc.flatten_frame_descriptors(h).numpy()                 # [3, 75000]
    s = dlc.match(frame_desc, frame_desc)                                 # cosine 3x3 (bf16 storage)
    so = ocos.scores(ocos.l2_normalize(frame_desc), ocos.l2_normalize(frame_desc))
    assert s.shape == (3, 3) and np.abs(s - so).max() < 5e-3              # bf16 rounding of the stored rows
    ts, ti = dlc.match_topk(frame_desc, frame_desc, 2)
    assert ti[:, 0].tolist() == [0, 1, 2]
    m = dlc.SimilarityCalculator(h.reshape(3, 30, 2500)).similarity_matrix()
    assert np.array_equal(m, osim.similarity_matrix(ho.reshape(3, 30, 2500)))


@pytest.mark.gpu
def test_drivers_and_database_file(tmp_path):
    import deeploopcloser_amd as dlc
    from deeploopcloser_amd import drivers
    frames_dir = os.path.join(GOLDEN, "frames")
    sim = drivers.create_similarity_matrix(frames_dir, out_png=str(tmp_path / "sim.png"), network=dlc.SDAV(seed=4))
    assert sim.shape == (3, 3) and sim.dtype == np.int64 and np.all(np.diag(sim) == -1) and np.array_equal(sim, sim.T)
    dist = drivers.create_distance_matrix(frames_dir, out_png=str(tmp_path / "dist.png"))
    assert dist.shape == (3, 3) and np.all(np.diag(dist) == 0) and np.array_equal(dist, dist.T) and dist.max() > 0
    from PIL import Image
    assert Image.open(tmp_path / "sim.png").size == (3, 3) and Image.open(tmp_path / "dist.png").size == (3, 3)
    img = drivers.distance_image(dist)
    assert img.max() == 255 and img.min() == 0
    # key-frame DB shard round trip
    rng = np.random.RandomState(0)
    db = dlc.KeyframeDatabase(rng.standard_normal((500, 100)).astype(np.float32), dtype="f16", center=True, row_offset=77)
    db.save(str(tmp_path / "shard.npz"))
    db2 = dlc.KeyframeDatabase.load(str(tmp_path / "shard.npz"))
    assert torch.equal(db.rows, db2.rows) and db2.row_offset == 77 and db2.center and db2.dtype == torch.float16
    q = rng.standard_normal((5, 100)).astype(np.float32)
    a, b = db.match_topk(q, 3), db2.match_topk(q, 3)
    assert torch.equal(a[1], b[1]) and torch.equal(a[0], b[0])


@pytest.mark.gpu
def test_rgb_to_gray_odd_sizes_and_unaligned_views():
    """The grey kernel handles four pixels per thread from whole dwords: pixel counts that are not multiples of four
    (the tail) and buffers that do not start on a dword (the byte path) give the same bytes as the oracle."""
    import deeploopcloser_amd as dlc
    eng = dlc.default_engine()
    rng = np.random.RandomState(12)
    for h, w in [(1, 1), (3, 5), (7, 9), (64, 66), (33, 127)]:
        rgb = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
        want = opatch.bgr2gray_opencv(rgb)
        assert np.array_equal(eng.rgb_to_gray(torch.from_numpy(rgb).cuda()).cpu().numpy(), want), (h, w)
        flat = torch.zeros((h * w * 3 + 3,), dtype=torch.uint8, device=eng.device)
        flat[3:] = torch.from_numpy(rgb).cuda().reshape(-1)                   # a view one pixel (3 bytes) into the buffer
        view = flat[3:].view(h, w, 3)
        assert view.data_ptr() % 4 != 0 and view.is_contiguous()
        assert np.array_equal(eng.rgb_to_gray(view).cpu().numpy(), want), (h, w)


@pytest.mark.gpu
def test_harris_keypoints_bit_exact_vs_oracle():
    """The build's detector (stand-in for SURF) is integer arithmetic: points, responses and counts
    equal the oracle's exactly -- reference frames, noise, a symmetric pattern full of ties, flat."""
    import deeploopcloser_amd as dlc
    from oracle import keypoints as okp
    eng = dlc.default_engine()
    rng = np.random.RandomState(8)
    frames = [opatch.bgr2gray_opencv(opatch.read_ppm(p)) for p in FRAMES]
    frames.append(rng.randint(0, 256, (192, 240)).astype(np.uint8))
    tie = np.zeros((192, 240), dtype=np.uint8)
    tie[::16, :] = 255
    tie[:, ::16] = 255                                                       # a lattice: hundreds of equal corners
    frames.append(tie)
    frames.append(np.full((192, 240), 90, dtype=np.uint8))                    # flat: no key-point
    sq = np.zeros((192, 240), dtype=np.uint8)
    sq[50:90, 60:140] = 180                                                   # 4 corners only: count < n
    frames.append(sq)
    dots = np.zeros((192, 240), dtype=np.uint8)
    for a in range(6):
        for b in range(8):                                                    # one blob per 32 x 32 tile of the kernel: the
            dots[14 + 32 * a:18 + 32 * a, 14 + 32 * b:18 + 32 * b] = 40 + 4 * (a * 8 + b)   # lists' first slots hold everything
    frames.append(dots)
    g = torch.from_numpy(np.stack(frames)).to(eng.device)
    for n in (30, 7, 200):
        pts, resp, cnt = eng.harris_keypoints(g, n)
        for f, img in enumerate(frames):
            ep, er, ec = okp.key_points(img, n)
            assert int(cnt[f]) == ec
            assert np.array_equal(pts[f].cpu().numpy(), ep) and np.array_equal(resp[f].cpu().numpy(), er)
    assert int(cnt[5]) == 0 and 0 < int(eng.harris_keypoints(g, 30)[2][6]) < 30
    with pytest.raises(ValueError):
        eng.harris_keypoints(g[:, :5, :5].contiguous(), 3)


@pytest.mark.gpu
def test_frontend_fuzz_odd_image_sizes():
    """Detector and patch gather on odd frame sizes (down to the 7x7 / 41x41 minima), seeded."""
    import deeploopcloser_amd as dlc
    from oracle import keypoints as okp
    eng = dlc.default_engine()
    rng = np.random.RandomState(31)
    for h, w, n in [(7, 7, 3), (8, 300, 5), (41, 41, 30), (100, 57, 30), (193, 241, 64), (96, 160, 300), (80, 72, 700)]:
        imgs = rng.randint(0, 256, (3, h, w)).astype(np.uint8)
        pts, resp, cnt = eng.harris_keypoints(torch.from_numpy(imgs).to(eng.device), n)
        for f in range(3):
            ep, er, ec = okp.key_points(imgs[f], n)
            assert int(cnt[f]) == ec and np.array_equal(pts[f].cpu().numpy(), ep) and np.array_equal(resp[f].cpu().numpy(), er), (h, w)
        if h >= 41 and w >= 41:
            parser = dlc.CvInputParser(n, 41)
            kps = dlc.harris_key_points(imgs[0], n)
            assert np.array_equal(parser.parse(imgs[0], kps), opatch.parse(imgs[0], [k.pt for k in kps], 41)), (h, w)


@pytest.mark.gpu
def test_parser_default_detector_and_batch():
    """parse(image) without key-points = Harris key-points through the reference's patch rule;
    parse_batch does the same for a stack of frames on the GPU, topping up with grid points."""
    import deeploopcloser_amd as dlc
    parser = dlc.CvInputParser(30, 41)
    rgb = [opatch.read_ppm(p) for p in FRAMES]
    batch = parser.parse_batch(np.stack(rgb)).cpu().numpy()
    for f, img in enumerate(rgb):
        kps = dlc.harris_key_points(img, 30)
        assert len(kps) == 30 and all(a.response >= b.response for a, b in zip(kps, kps[1:]))
        one = parser.parse(img)
        assert one.shape == (30, 1681) and np.array_equal(one, parser.parse(img, kps)) and np.array_equal(one, batch[f])
        assert np.array_equal(one, opatch.parse(opatch.bgr2gray_opencv(img), [kp.pt for kp in kps], 41))
    sq = np.zeros((192, 240), dtype=np.uint8)
    sq[50:90, 60:140] = 180
    few = dlc.harris_key_points(sq, 30)
    x = parser.parse(sq)
    grid = dlc.grid_key_points((192, 240), 30)
    assert 0 < len(few) < 30 and x.shape == (30, 1681)
    assert np.array_equal(x[:len(few)], parser.parse(sq, few))
    assert np.array_equal(x[len(few):], parser.parse(sq, grid)[len(few):])
