"""The 20 real frames of the reference's datasets/test (tests/golden/frames + tests/golden/datasets_test: data files) tiled
to any number of frames -- real-image statistics at BASELINE configs[1]'s full size (the reference's
outdoor_kennedylong is 1063 such frames; the repo carries 20).  Used by tests/test_gpu_fullsize.py and by bench.py's
`paths` rows; no oracle import (bench.py's product legs load this)."""
import glob
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def frame_paths():
    paths = glob.glob(os.path.join(GOLDEN, "frames", "*.ppm")) + glob.glob(os.path.join(GOLDEN, "datasets_test", "*.ppm"))
    return sorted(paths, key=os.path.basename)


def tiled_patches(dlc, n_frames):
    """[n_frames, 30, 1681] fp64 on the device: the real frames through the GPU front-end (grey, Harris, patches), then copies:
    copies 0 and 1 exact (identical descriptors: +inf scores, every arg-min a tie of bit-identical rows), later ones with a
    few pixels moved by 1/255, every fifth one also with blank patches and a key-point found twice."""
    import torch
    parser = dlc.CvInputParser(30, 41)
    x = parser.parse_batch(np.stack([dlc.read_ppm(p) for p in frame_paths()]))     # [20, 30, 1681] on the device
    rng = np.random.RandomState(8)
    tiles, have, c = [x], x.shape[0], 0
    while have < n_frames:
        t = x.clone()
        if c >= 2:
            for f in range(x.shape[0]):
                for _ in range(1 + c % 10):
                    t[f, rng.randint(30), rng.randint(1681)] += (1.0 if rng.rand() < 0.5 else -1.0) / 255.0
            t.clamp_(0.0, 1.0)
        if c >= 6 and c % 5 == 1:
            t[:, 3] = 0.0; t[:, 4] = 0.0; t[::2, 9] = t[::2, 8]                    # blank patches, a key-point found twice
        tiles.append(t)
        have += t.shape[0]
        c += 1
    return torch.cat(tiles)[:n_frames]


def tiled_bgr_frames(dlc, n_frames):
    """[n_frames, 192, 240, 3] float64 on the device, as create_distance_matrix.py:23 feeds CnnVtl (cv2.imread order: BGR,
    uint8 values): the 20 real frames, then copies -- copies 0 and 1 exact, later ones with a handful of pixel values moved
    by +-1 (clamped to 0..255), every seventh one also with a blanked 32 x 32 block."""
    import torch
    base = np.stack([dlc.read_ppm(p)[:, :, ::-1] for p in frame_paths()]).astype(np.float64)      # [20, 192, 240, 3]
    x = torch.from_numpy(np.ascontiguousarray(base)).to("cuda")
    rng = np.random.RandomState(18)
    tiles, have, c = [x], x.shape[0], 0
    while have < n_frames:
        t = x.clone()
        if c >= 2:
            for f in range(x.shape[0]):
                for _ in range(3 + c % 7):
                    t[f, rng.randint(192), rng.randint(240), rng.randint(3)] += 1.0 if rng.rand() < 0.5 else -1.0
            t.clamp_(0.0, 255.0)
        if c >= 4 and c % 7 == 3:
            y0, x0 = rng.randint(160), rng.randint(208)
            t[:, y0:y0 + 32, x0:x0 + 32] = 0.0
        tiles.append(t)
        have += t.shape[0]
        c += 1
    return torch.cat(tiles)[:n_frames]


def tiled_u8_frames(dlc, n_frames):
    """[n_frames, 192, 240, 3] uint8 RGB on the HOST -- what a script reads from disk before anything runs
    (create_similarity_matrix.py:23-26, create_distance_matrix.py:20-25; cv2.imread's BGR is this array's [..., ::-1]):
    the 20 real frames, then copies -- copies 0 and 1 exact, later ones with a handful of pixel values moved by +-1,
    every seventh one also with a blanked 32 x 32 block."""
    base = np.stack([dlc.read_ppm(p) for p in frame_paths()])                                      # [20, 192, 240, 3]
    rng = np.random.RandomState(28)
    tiles, have, c = [base], base.shape[0], 0
    while have < n_frames:
        t = base.astype(np.int16)
        if c >= 2:
            for f in range(base.shape[0]):
                for _ in range(3 + c % 7):
                    t[f, rng.randint(192), rng.randint(240), rng.randint(3)] += 1 if rng.rand() < 0.5 else -1
            np.clip(t, 0, 255, out=t)
        if c >= 4 and c % 7 == 3:
            y0, x0 = rng.randint(160), rng.randint(208)
            t[:, y0:y0 + 32, x0:x0 + 32] = 0
        tiles.append(t.astype(np.uint8))
        have += base.shape[0]
        c += 1
    return np.ascontiguousarray(np.concatenate(tiles)[:n_frames])
