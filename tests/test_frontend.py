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
def test_end_to_end_pipelines_equal_the_staged_calls(tmp_path):
    """BASELINE configs[1] / configs[2] as one device-resident path (deeploopcloser_amd/pipeline.py; the reference's scripts
    src/sdav/create_similarity_matrix.py:23-38 and src/cnn_vtl/create_distance_matrix.py:14-36): frames in, matrix out, no
    host hop between the stages.  On the 20 real frames tiled to 47 (three upload chunks of 16): the matrices equal, bit for
    bit, what the staged calls (parser -> transform -> matrix, each through NumPy) give; host frames == resident frames;
    caller-supplied key-points == the per-frame parse; the fp64 similarity equals the oracle's on the first frames."""
    import deeploopcloser_amd as dlc
    from deeploopcloser_amd import pipeline, drivers
    from oracle import similarity as osim
    import real_frames
    paths = real_frames.frame_paths()
    base = np.stack([dlc.read_ppm(p) for p in paths])                          # [20, 192, 240, 3] uint8 RGB
    rng = np.random.RandomState(3)
    frames = np.concatenate([base, base[::-1], base[:7]])                      # 47 frames
    noise = rng.randint(0, 40, size=frames[20:].shape)
    frames[20:] = np.clip(frames[20:].astype(np.int64) + noise - 20, 0, 255).astype(np.uint8)
    n = frames.shape[0]
    net = dlc.SDAV(seed=4, weight_scale="fan_in")
    parser = dlc.CvInputParser(30, 41)
    # staged, through the host between every stage (what drivers.py did before)
    x = np.stack([parser.parse(fr) for fr in frames])
    h = net.transform(x)
    want = dlc.SimilarityCalculator(h.reshape(n, 30, 2500)).similarity_matrix()
    t = []
    got = pipeline.sdav_similarity_matrix_from_frames(frames, net, parser, chunk_frames=16, timings=t)
    assert got.dtype == np.int64 and np.array_equal(got, want)
    ms = pipeline.stage_ms(t)
    assert set(ms) == {"front-end (grey, key-points, patches)", "SDAV.transform",
                       "similarity matrix (distinctive score + all-vs-all)", "download of the matrix"} and all(v > 0 for v in ms.values())
    res = pipeline.sdav_similarity_matrix_from_frames(torch.from_numpy(frames).cuda(), net, device_result=True)
    assert res.is_cuda and np.array_equal(res.cpu().numpy(), want)
    f64 = pipeline.sdav_similarity_matrix_from_frames(frames, net, as_int64=False)
    hd = h.reshape(n, 30, 2500)
    for i_, j_ in ((0, 1), (2, 5), (3, 40), (21, 46)):                         # (the dataset's mean is over all 47 frames)
        ref = osim.similarity_score(hd, hd[i_], hd[j_])
        assert abs(f64[i_, j_] - ref) <= 1e-9 * abs(ref) and f64[j_, i_] == f64[i_, j_]
    # the tolerance mode's encoder in the same pipeline: its own staged result
    net16 = dlc.SDAV(seed=4, weight_scale="fan_in", dtype="f16x2")
    want16 = dlc.SimilarityCalculator(net16.transform(x).reshape(n, 30, 2500)).similarity_matrix()
    assert np.array_equal(pipeline.sdav_similarity_matrix_from_frames(frames, net16, chunk_frames=20), want16)
    # caller-supplied key-points (the reference's SURF slot): per-frame parse == the batched gather
    kps = [dlc.grid_key_points(fr.shape[:2], 30) for fr in frames[:5]]
    xk = np.stack([parser.parse(fr, kp) for fr, kp in zip(frames[:5], kps)])
    wantk = dlc.SimilarityCalculator(net.transform(xk).reshape(5, 30, 2500)).similarity_matrix()
    gotk = pipeline.sdav_similarity_matrix_from_frames(frames[:5], net, parser, key_points=pipeline.key_point_array(kps, 30, net.engine))
    assert np.array_equal(gotk, wantk)
    with pytest.raises(ValueError, match="key-points"):
        pipeline.key_point_array([kps[0][:7]], 30, net.engine)
    # configs[2]
    bgr = np.ascontiguousarray(frames[..., ::-1])
    cnn = dlc.CnnVtl(input_shape=[n, 192, 240, 3], seed=3, mask_seed=4)
    d8 = cnn.transform(bgr)
    wantd = dlc.DistanceCalculator.distance_matrix(d8)
    t = []
    gotd = pipeline.cnn_vtl_distance_matrix_from_frames(bgr, cnn, chunk_frames=16, timings=t)
    assert gotd.dtype == np.int64 and np.array_equal(gotd, wantd) and set(pipeline.stage_ms(t)) == {
        "CnnVtl.transform", "distance matrix", "download of the matrix"}
    assert np.array_equal(pipeline.cnn_vtl_distance_matrix_from_frames(torch.from_numpy(bgr).cuda(), cnn, device_result=True).cpu().numpy(), wantd)
    assert np.array_equal(pipeline.cnn_vtl_descriptors_from_frames(bgr.astype(np.float64), cnn).cpu().numpy(), d8)
    # the drivers are these pipelines behind a directory listing (also with a key-point function)
    frames_dir = os.path.join(GOLDEN, "frames")
    files = sorted(os.listdir(frames_dir))
    three = np.stack([dlc.read_ppm(os.path.join(frames_dir, f)) for f in files])
    sim = drivers.create_similarity_matrix(frames_dir, network=net)
    assert np.array_equal(sim, pipeline.sdav_similarity_matrix_from_frames(three, net))
    simk = drivers.create_similarity_matrix(frames_dir, network=net, key_points_fn=lambda shape: dlc.grid_key_points(shape, 30))
    xg = np.stack([parser.parse(fr, dlc.grid_key_points(fr.shape[:2], 30)) for fr in three])
    assert np.array_equal(simk, dlc.SimilarityCalculator(net.transform(xg).reshape(3, 30, 2500)).similarity_matrix())


@pytest.mark.gpu
def test_pipeline_edge_cases():
    """pipeline.py on the inputs either side of the usual one: no frames, one frame, grey frames, a chunk larger than the
    batch, a first chunk shorter than the others (Engine.for_each_chunk), resident float64 frames for CnnVtl."""
    import deeploopcloser_amd as dlc
    from deeploopcloser_amd import pipeline
    import real_frames
    eng = dlc.default_engine()
    frames = np.stack([dlc.read_ppm(p) for p in real_frames.frame_paths()[:9]])
    net = dlc.SDAV(seed=2, weight_scale="fan_in")
    parser = dlc.CvInputParser(30, 41)
    want = pipeline.sdav_descriptors_from_frames(torch.from_numpy(frames).cuda(), net, parser)
    assert want.shape == (9, 30, 2500) and want.is_cuda
    assert pipeline.sdav_descriptors_from_frames(frames[:0], net, parser).shape == (0, 30, 2500)
    assert torch.equal(pipeline.sdav_descriptors_from_frames(frames[:1], net, parser), want[:1])
    for chunk, first in ((100, None), (4, None), (4, 1), (2, 2), (5, 3)):
        pipeline.FIRST_CHUNK_FRAMES = first
        try:
            assert torch.equal(pipeline.sdav_descriptors_from_frames(frames, net, parser, chunk_frames=chunk), want), (chunk, first)
        finally:
            pipeline.FIRST_CHUNK_FRAMES = None
    # grey frames in == the grey conversion of the colour frames in
    gray = eng.rgb_to_gray(torch.from_numpy(frames).cuda()).cpu().numpy()
    assert torch.equal(pipeline.sdav_descriptors_from_frames(gray, net, parser), want)
    # the chunk walker itself: every item visited once, in order, whatever the chunking
    x = np.arange(23 * 5, dtype=np.int32).reshape(23, 5)
    for chunk, first in ((23, None), (7, None), (7, 2), (1, None), (50, 3)):
        seen = []
        eng.for_each_chunk(x, chunk, lambda dev, lo, hi: seen.append((lo, hi, dev.clone())), first=first)
        torch.cuda.synchronize()
        assert [s_[0] for s_ in seen] == [0] + [s_[1] for s_ in seen[:-1]] and seen[-1][1] == 23
        assert np.array_equal(torch.cat([s_[2] for s_ in seen]).cpu().numpy(), x)
    cnn = dlc.CnnVtl(input_shape=[9, 192, 240, 3], seed=3, mask_seed=4)
    bgr = np.ascontiguousarray(frames[..., ::-1])
    d8 = pipeline.cnn_vtl_descriptors_from_frames(bgr, cnn)
    assert torch.equal(pipeline.cnn_vtl_descriptors_from_frames(torch.from_numpy(bgr).cuda().to(torch.float64), cnn), d8)
    assert pipeline.cnn_vtl_descriptors_from_frames(bgr[:0], cnn).shape == (0, cnn.columns.size)
    assert pipeline.cnn_vtl_distance_matrix_from_frames(bgr[:1], cnn).tolist() == [[0]]


@pytest.mark.gpu
def test_cli_reports_per_frame_latency(capsys):
    """The streaming CLI one frame at a time: candidates on stdout, the step latency (file -> descriptors -> match ->
    candidates on the host) on stderr."""
    from deeploopcloser_amd import loop_closure
    rc = loop_closure.main([os.path.join(GOLDEN, "datasets_test"), "--network", "cnn_vtl", "--k", "1", "--exclusion", "0",
                            "--threshold", "-1", "--batch", "1"])
    cap = capsys.readouterr()
    assert rc == 0 and len(cap.out.strip().splitlines()) == 16              # 17 frames: all but the first find an older one
    line = [l for l in cap.err.splitlines() if l.startswith("latency\t")]
    assert len(line) == 1
    f = line[0].split("\t")
    assert f[1:5] == ["batch", "1", "steps", "17"] and float(f[f.index("ms_per_frame") + 1]) > 0.0


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
