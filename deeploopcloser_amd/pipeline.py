"""BASELINE configs[1] / configs[2] as ONE device-resident path each: frames in, matrix out.

The reference's user runs a script -- src/sdav/create_similarity_matrix.py:23-38 (frames -> CvInputParser -> SDAV.transform
-> the all-vs-all SimilarityCalculator loop) or src/cnn_vtl/create_distance_matrix.py:14-36 (frames -> CnnVtl.transform ->
the all-vs-all DistanceCalculator loop).  Every stage of those scripts exists here as a kernel; this module composes them
WITHOUT a host hop between stages: one upload of the uint8 frames (in chunks that overlap the first kernels), descriptors
that never leave HBM, one download of the N x N matrix.  drivers.py and the streaming CLI (loop_closure.py) are built on it.

Stage times: pass timings=[] and read stage_ms(timings) after the call -- HIP events on the launch stream around every
stage of every chunk (the stages of different chunks interleave; a stage's figure is the sum over the chunks).
"""
import numpy as np
import torch

from .input import CvInputParser, _centres


RESIDENT_CHUNK_FRAMES = 2048
FIRST_CHUNK_FRAMES = None        # frames of the first upload chunk (None: like the others).  The first upload overlaps nothing, but a
                                 # short first chunk did not pay: its kernels fill a fraction of the chip (scripts/exp_first_chunk.py:
                                 # 39.1 / 16.8 / 32.6 ms host to host with None, 39.4-40.5 / 16.8-17.5 / 32.2-32.8 with 16 .. 128 frames)


def _mark(timings, name, stream):
    if timings is None:
        return None
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    return (name, e0)


def _done(timings, mark, stream):
    if mark is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record(stream)
        timings.append((mark[0], mark[1], e1))


def stage_ms(timings):
    """{stage: milliseconds summed over the chunks} of a finished call (synchronises on the recorded events)."""
    out = {}
    for name, e0, e1 in timings:
        e1.synchronize()
        out[name] = out.get(name, 0.0) + e0.elapsed_time(e1)
    return out


def _patches(parser, dev_frames, key_points, lo, hi):
    """uint8 frames of one chunk on the device -> [B, P, patch^2] fp64 (grey, key-points, gather: CvInputParser.py:19-49)."""
    if key_points is None:
        return parser.parse_batch(dev_frames)
    e = parser.engine
    gray = e.rgb_to_gray(dev_frames) if dev_frames.dim() == 4 else dev_frames
    return e.extract_patches(gray, key_points[lo:hi], parser.patch_size)


def key_point_array(key_points_per_frame, n_patches, engine):
    """The caller's key-points (per frame: objects with .pt [and .response], or (x, y) pairs -- CvInputParser.py:36-46) as ONE
    int32 device tensor [N, n_patches, 2]; every frame must supply at least n_patches (the network needs the shape; the
    reference's np.array of ragged parses fails in TensorFlow)."""
    rows = []
    for f, kps in enumerate(key_points_per_frame):
        c = _centres(kps, n_patches)
        if c.shape[0] != n_patches:
            raise ValueError("frame %d: %d key-points, the network needs %d patches per frame" % (f, c.shape[0], n_patches))
        rows.append(c)
    return torch.from_numpy(np.stack(rows)).to(engine.device)


def sdav_descriptors_from_frames(frames, network, parser=None, key_points=None, chunk_frames=256, timings=None):
    """frames: uint8 [N, H, W, 3] RGB (or [N, H, W] grey), a host ndarray or a device tensor -> the frames' SDAV descriptors
    [N, P, H_last] fp64 ON THE DEVICE (CvInputParser.parse + SDAV.transform of create_similarity_matrix.py:23-27, batched).
    key_points: None (the build's Harris detector) or an int32 device tensor [N, P, 2] (key_point_array).
    A host array is uploaded in chunks of `chunk_frames` frames whose transfer overlaps the previous chunk's kernels; the
    encoder is batch-invariant, so chunking changes no bit."""
    eng = network.engine
    parser = parser or CvInputParser(network.input_shape[0], int(round(np.sqrt(network.input_shape[1]))), device=eng.device)
    p, hw = network.input_shape[0], network.hidden_units[-1]
    n = int(frames.shape[0])
    desc = torch.empty((n, p, hw), dtype=torch.float64, device=eng.device)
    cur = torch.cuda.current_stream(eng.device)

    def consume(dev, lo, hi):
        m = _mark(timings, "front-end (grey, key-points, patches)", cur)
        x = _patches(parser, dev, key_points, lo, hi)
        _done(timings, m, cur)
        m = _mark(timings, "SDAV.transform", cur)
        if network.dtype == torch.float64:                      # straight into its place (behind the call the copy was 638 MB
            network.transform_tensor(x, out=desc[lo:hi].view(-1, hw))   # read and written again: 0.25 ms per 1063 frames)
        else:
            desc[lo:hi] = network.transform_tensor(x).view(hi - lo, p, hw)
        _done(timings, m, cur)

    if n == 0:
        return desc
    if isinstance(frames, torch.Tensor):
        dev = frames.to(eng.device)
        if dev.dtype != torch.uint8:
            dev = dev.to(torch.uint8)
        step = RESIDENT_CHUNK_FRAMES                       # resident frames: nothing to overlap, chunks only bound the
        for lo in range(0, n, step):                       # activations' footprint (2.6 GB per 2048 frames)
            consume(dev[lo:lo + step].contiguous(), lo, min(n, lo + step))
    else:
        a = np.asarray(frames)
        if a.dtype != np.uint8:
            a = a.astype(np.uint8)
        n_chunks = max(1, -(-n // max(1, int(chunk_frames))))
        eng.for_each_chunk(a, -(-n // n_chunks), consume, first=FIRST_CHUNK_FRAMES)
    return desc


def sdav_similarity_matrix_from_frames(frames, network, parser=None, key_points=None, chunk_frames=256, mu=0.5, sigma=0.2,
                                       a=10, b=-10, as_int64=True, device_result=False, timings=None):
    """create_similarity_matrix.py:23-38 end to end: frames -> patches -> SDAV descriptors -> the all-vs-all similarity matrix
    (int64 as the reference stores it: truncated scores, upper triangle mirrored, diagonal -1; as_int64=False: the fp64
    scores).  One upload, one download (device_result=True: none -- the matrix stays a device tensor)."""
    eng = network.engine
    cur = torch.cuda.current_stream(eng.device)
    desc = sdav_descriptors_from_frames(frames, network, parser, key_points, chunk_frames, timings)
    m = _mark(timings, "similarity matrix (distinctive score + all-vs-all)", cur)
    score, rng = eng.distinctive_score(desc, mu, sigma, with_range=True)
    f, i = eng.sdav_similarity_matrix(desc, score, a, b, want_int64=as_int64, range=rng)
    _done(timings, m, cur)
    out = i if as_int64 else f
    if device_result:
        return out
    m = _mark(timings, "download of the matrix", cur)
    res = eng.download(out)
    _done(timings, m, cur)
    return res


def cnn_vtl_descriptors_from_frames(frames, network, chunk_frames=None, timings=None):
    """frames: [N, H, W, 3] BGR as cv2.imread returns them (uint8; float64 accepted), host ndarray or device tensor -> int8
    descriptors [N, D'] ON THE DEVICE (CnnVtl.transform, cnn_vtl.py:130-133).  Host frames upload in chunks that overlap
    the previous chunk's convolutions; a frame's descriptor does not depend on its chunk."""
    eng = network.engine
    n = int(frames.shape[0])
    out = torch.empty((n, network.columns.size), dtype=torch.int8, device=eng.device)
    cur = torch.cuda.current_stream(eng.device)

    def consume(dev, lo, hi):
        m = _mark(timings, "CnnVtl.transform", cur)
        out[lo:hi] = network.transform_tensor(dev)
        _done(timings, m, cur)

    if n == 0:
        return out
    if isinstance(frames, torch.Tensor):
        consume(frames, 0, n)                                  # (transform_tensor chunks a large batch itself)
        return out
    a = np.asarray(frames)
    if a.dtype not in (np.uint8, np.float64, np.float32):
        a = a.astype(np.float64)
    cf = int(chunk_frames or max(1, network.frame_chunk // 4))
    n_chunks = max(1, -(-n // cf))
    eng.for_each_chunk(a, -(-n // n_chunks), consume, first=FIRST_CHUNK_FRAMES)
    return out


def cnn_vtl_distance_matrix_from_frames(frames, network, chunk_frames=None, device_result=False, timings=None):
    """create_distance_matrix.py:14-36 end to end: frames -> CnnVtl int8 descriptors -> the full N x N int64 matrix of
    DistanceCalculator.calculate_distance.  One upload, one download (device_result=True: none)."""
    eng = network.engine
    cur = torch.cuda.current_stream(eng.device)
    d8 = cnn_vtl_descriptors_from_frames(frames, network, chunk_frames, timings)
    m = _mark(timings, "distance matrix", cur)
    dm = eng.cnnvtl_distance_matrix(d8)
    _done(timings, m, cur)
    if device_result:
        return dm
    m = _mark(timings, "download of the matrix", cur)
    res = eng.download(dm)
    _done(timings, m, cur)
    return res
