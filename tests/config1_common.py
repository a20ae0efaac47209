"""BASELINE configs[0] at its stated size: the 20 frames of the reference's datasets/test
(tests/golden/frames/ holds 000000-000002, tests/golden/datasets_test/ 000003-000019 -- data files),
taken through the CPU oracle end to end.  Shared by tests/test_config1.py and the script that captured
the reference-generated golden matrix (tests/golden/make_config1_golden.py).  Test infrastructure only."""
import glob
import os

import numpy as np

from oracle import cosine as ocos
from oracle import keypoints as okp
from oracle import patches as opatch
from oracle import sdav as osdav

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
N_PATCHES, PATCH = 30, 41
SEED = 4


def frame_paths():
    """datasets/test, sorted by name (create_distance_matrix.py:15-16 sorts; InputGenerator.py:17 does not)."""
    paths = glob.glob(os.path.join(GOLDEN, "frames", "*.ppm")) + glob.glob(os.path.join(GOLDEN, "datasets_test", "*.ppm"))
    return sorted(paths, key=os.path.basename)


def grid_points(shape, n):
    """The build's deterministic top-up centres (deeploopcloser_amd.input.grid_key_points restated)."""
    rows = int(np.ceil(np.sqrt(n * shape[0] / shape[1])))
    cols = int(np.ceil(n / rows))
    xs = (np.arange(rows) + 0.5) * shape[0] / rows
    ys = (np.arange(cols) + 0.5) * shape[1] / cols
    return [(float(x), float(y)) for x in xs for y in ys][:n]


def oracle_patches(paths):
    """frames -> grey (cv2.imread's fixed-point BT.601) -> the 30 strongest Harris corners (the build's
    stand-in for SURF, oracle/keypoints.py; fewer corners: topped up with grid points) -> 41x41 patches
    with the reference's window rule (CvInputParser.py:19-33,49-123): [N, 30, 1681] in [0, 1]."""
    out = []
    for p in paths:
        gray = opatch.bgr2gray_opencv(opatch.read_ppm(p))
        pts, _, count = okp.key_points(gray, N_PATCHES)
        centres = [(float(x), float(y)) for x, y in pts[:count]]
        centres += grid_points(gray.shape, N_PATCHES)[count:]
        out.append(opatch.parse(gray, centres, PATCH))
    return np.stack(out)


def oracle_descriptors(x, scale):
    """SDAV.transform (SDAV.py:293-302) with seeded weights: the reference's N(0,1) initialiser
    (scale='reference') or the 1/sqrt(fan_in) regime."""
    ws, bs = osdav.init_weights(SEED, scale=scale)
    return osdav.transform(x, ws, bs)                                    # [N*30, 2500]


def oracle_cosine(h, n):
    """The N x N cosine matrix of the flattened [30*2500] place descriptors (mean-centred), fp64."""
    place = ocos.l2_normalize(h.reshape(n, -1), center=True)
    return ocos.scores(place, place)


def bgr_frames(paths):
    """The frames as create_distance_matrix.py:23 feeds them: cv2.imread order (BGR), uint8 values as float64."""
    return np.stack([opatch.read_ppm(p)[:, :, ::-1] for p in paths]).astype(np.float64)


def oracle_cnn_descriptors(paths, seed=3, mask_seed=4):
    """CnnVtl.transform (cnn_vtl.py:28-133) of the frames with seeded AlexNet-shaped weights and column mask."""
    from oracle import cnn_vtl as ocnn
    x = bgr_frames(paths)
    ws, bs = ocnn.init_weights(seed)
    cols = ocnn.column_indices(ocnn.layer_sizes(x.shape[1:3]), 99.59, seed=mask_seed)
    return ocnn.transform(x, ws, bs, cols)
