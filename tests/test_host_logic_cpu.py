"""Host-side logic of the Python mirror that runs without a GPU."""
import glob
import os
from collections import namedtuple

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import patches as opatch

FRAMES = sorted(glob.glob(os.path.join(GOLDEN, "frames", "*.ppm")))


def test_read_ppm_matches_oracle_reader():
    from deeploopcloser_amd.input import read_ppm
    for f in FRAMES:
        a = read_ppm(f)
        assert a.shape == (192, 240, 3) and np.array_equal(a, opatch.read_ppm(f))
    with pytest.raises(ValueError):
        read_ppm(os.path.join(GOLDEN, "similarity.npz"))


def test_key_point_ordering_and_rounding():
    """CvInputParser.py:45-46 (descending response, top n) and :111 (Python round, half to even)."""
    from deeploopcloser_amd.input import _centres, grid_key_points
    KP = namedtuple("KP", "pt response")
    kps = [KP((10.5, 20.5), 0.1), KP((11.5, 21.5), 0.9), KP((0.49, 239.51), 0.5), KP((5, 5), 0.7)]
    c = _centres(kps, 3)
    assert c.tolist() == [[12, 22], [5, 5], [0, 240]] and c.dtype == np.int32
    assert _centres([(1.5, 2.5), (3.5, 4.5)], 30).tolist() == [[2, 2], [4, 4]]
    assert _centres([], 30).shape == (0, 2)
    g = grid_key_points((192, 240), 30)
    assert len(g) == 30 and len(set(g)) == 30
    assert all(0 <= x < 192 and 0 <= y < 240 for x, y in g)


def test_driver_image_formulas():
    """create_similarity_matrix.py:41-45 and create_distance_matrix.py:40."""
    from deeploopcloser_amd.drivers import similarity_image, distance_image
    m = np.array([[-1, 30, 10], [30, -1, 50], [10, 50, -1]], dtype=np.int64)
    img = similarity_image(m)
    move = 0 - m.min()
    assert np.allclose(img, 255 * ((m + move) / (m.max() + move))) and img.min() == 0 and img.max() == 255
    d = np.array([[0, 5], [5, 0]], dtype=np.int64)
    assert np.array_equal(distance_image(d), np.array([[255.0, 0.0], [0.0, 255.0]]))


def test_flatten_frame_descriptors_and_dtype_names():
    import torch
    from deeploopcloser_amd.matching import flatten_frame_descriptors
    from deeploopcloser_amd.engine import torch_dtype
    h = np.arange(2 * 30 * 4, dtype=np.float64).reshape(60, 4)
    f = flatten_frame_descriptors(h)
    assert tuple(f.shape) == (2, 120) and np.array_equal(f[1].numpy(), h[30:].reshape(-1))
    assert torch_dtype("bf16") == torch.bfloat16 and torch_dtype("fp16") == torch.float16
    with pytest.raises(ValueError):
        torch_dtype("int3")


def test_loop_closure_candidate_filter_and_oracle():
    """first_k_eligible keeps, per frame, the first k candidates that are old enough; the
    oracle's streaming rule on a hand-made sequence."""
    import torch
    from deeploopcloser_amd.loop_closure import first_k_eligible
    from oracle import loop_closure as oloop
    s = torch.tensor([[0.9, 0.8, 0.7, 0.6], [0.5, 0.4, float("-inf"), float("-inf")]])
    i = torch.tensor([[7, 2, 9, 1], [3, 0, -1, -1]])
    fs, fi = first_k_eligible(s, i, torch.tensor([8, 3]), 3)
    assert fi.tolist() == [[7, 2, 1], [0, -1, -1]]
    assert fs[0].tolist() == pytest.approx([0.9, 0.8, 0.6]) and fs[1, 0].item() == pytest.approx(0.4)
    assert torch.isneginf(fs[1, 1:]).all()
    fs, fi = first_k_eligible(s[:, :2], i[:, :2], torch.tensor([8, 3]), 3)       # fewer candidates than k
    assert fi.tolist() == [[7, 2, -1], [0, -1, -1]]

    e = np.eye(4)
    rows = np.stack([e[0], e[1], e[2], e[0], e[1], e[3], e[0]])            # places 0 1 2 0 1 3 0
    es, ei = oloop.stream_topk(rows, k=2, exclusion=1)
    assert ei[:2].tolist() == [[-1, -1], [-1, -1]]                          # nothing old enough yet
    assert ei[3].tolist() == [0, 1] and es[3].tolist() == [1.0, 0.0]        # frame 3 sees 0..1, revisits place 0
    assert ei[4, 0] == 1 and ei[6].tolist() == [0, 3]                       # ties -> lower id


def test_keypoint_oracle_on_a_square():
    """oracle/keypoints.py: the four strongest Harris corners of a bright square are its corners;
    mirror-symmetric responses tie and the lower row-major index comes first; a flat image has none."""
    from oracle import keypoints as okp
    img = np.zeros((40, 48), dtype=np.uint8)
    img[12:28, 16:36] = 200
    pts, resp, count = okp.key_points(img, 6)
    assert count >= 4 and resp[0] == resp[1] == resp[2] == resp[3] > 0
    corners = {(16, 12), (35, 12), (16, 27), (35, 27)}                       # (x = column, y = row)
    for x, y in pts[:4]:
        assert min(abs(x - cx) + abs(y - cy) for cx, cy in corners) <= 2
    lin = [int(y) * 48 + int(x) for x, y in pts[:4]]
    assert lin == sorted(lin)                                                # equal responses: ascending index
    assert okp.key_points(np.full((20, 20), 7, dtype=np.uint8), 5)[2] == 0
    assert okp.response(img)[:3].max() == 0 and okp.response(img)[:, -3:].max() == 0   # undefined margin is 0


def _grouped_alexnet_dict(rng, as_bytes=False):
    """A synthetic bvlc_alexnet.npy dict in the REAL layout: grouped kernels for conv2/4/5, fc layers present."""
    shapes = {"conv1": (11, 11, 3, 96), "conv2": (5, 5, 48, 256), "conv3": (3, 3, 256, 384),
              "conv4": (3, 3, 192, 384), "conv5": (3, 3, 192, 256), "fc6": (8, 4), "fc7": (4, 4), "fc8": (4, 2)}
    d = {}
    for name, shp in shapes.items():
        w = (rng.standard_normal(shp) / np.sqrt(np.prod(shp[:-1]))).astype(np.float32)
        b = (rng.standard_normal(shp[-1]) * 0.1).astype(np.float32)
        d[name.encode() if as_bytes else name] = [w, b]
    return d


def test_tf1_constant_fill_and_grouped_alexnet_dict():
    """load_alexnet_npy's host side (cnn_vtl.py:137-149): the blob's grouped kernels go through
    tf.constant_initializer into ungrouped variables -- C-order values, last value repeated."""
    from deeploopcloser_amd.cnn_vtl import tf1_constant_fill, alexnet_params_from_dict
    from oracle import cnn_vtl as ocnn
    # the documented TF-1 behaviour on a small case: value [0..7] into shape [2,3,2]
    got = tf1_constant_fill(np.arange(8), (2, 3, 2))
    assert np.array_equal(got.ravel(), [0, 1, 2, 3, 4, 5, 6, 7, 7, 7, 7, 7])
    assert np.array_equal(tf1_constant_fill(np.arange(6).reshape(3, 2), (2, 3)), np.arange(6).reshape(2, 3))
    with pytest.raises(ValueError):
        tf1_constant_fill(np.arange(7), (2, 3))
    for as_bytes in (False, True):
        d = _grouped_alexnet_dict(np.random.RandomState(5), as_bytes)
        ws, bs = alexnet_params_from_dict(d)
        ows, obs = ocnn.weights_from_alexnet_dict(d)
        assert [w.shape for w in ws] == [(11, 11, 3, 96), (5, 5, 96, 256), (3, 3, 256, 384), (3, 3, 384, 384),
                                         (3, 3, 384, 256)]
        for w, ow, b, ob in zip(ws, ows, bs, obs):
            assert w.dtype == np.float64 and np.array_equal(w, ow) and np.array_equal(b, ob)
        key = (lambda n: n.encode()) if as_bytes else (lambda n: n)
        g = np.asarray(d[key("conv2")][0], dtype=np.float64)
        assert np.array_equal(ws[1].ravel()[:g.size], g.ravel())            # first half: the blob in C order
        assert np.all(ws[1].ravel()[g.size:] == g.ravel()[-1])              # second half: its last value
        assert np.array_equal(ws[0], np.asarray(d[key("conv1")][0], dtype=np.float64))   # ungrouped layers unchanged
    with pytest.raises(KeyError):
        alexnet_params_from_dict({"conv1": d[key("conv1")]})


def test_space_to_depth_kernel_rearrangement():
    """CnnVtl runs conv1 (11x11, stride 4, VALID) as a stride-1 3x3 convolution over the space-to-depth(4) input: the
    rearranged kernel over the rearranged input gives the oracle's strided convolution (same products, zero taps)."""
    from deeploopcloser_amd.cnn_vtl import space_to_depth_kernel
    from oracle import cnn_vtl as ocnn
    rng = np.random.RandomState(0)
    for (h, w, c, k, s, o) in [(24, 32, 3, 11, 4, 5), (12, 12, 2, 3, 2, 4), (20, 28, 3, 7, 4, 3)]:
        x = rng.standard_normal((2, h, w, c))
        wk = rng.standard_normal((k, k, c, o))
        b = rng.standard_normal(o)
        ref = ocnn.conv2d_nhwc(x, wk, b, s, "VALID", True)
        xs = x.reshape(2, h // s, s, w // s, s, c).transpose(0, 1, 3, 2, 4, 5).reshape(2, h // s, w // s, s * s * c)
        wp = space_to_depth_kernel(wk, s)
        assert wp.shape == (-(-k // s), -(-k // s), s * s * c, o)
        got = ocnn.conv2d_nhwc(xs, wp, b, 1, "VALID", True)
        assert got.shape == ref.shape and np.abs(got - ref).max() < 1e-12


def test_streaming_cli_runs_as_a_module():
    """`python -m deeploopcloser_amd.loop_closure` is the documented streaming CLI (README, DESIGN, INTEGRATION): the module
    must end in a __main__ guard -- without one the command prints nothing and exits 0."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    res = subprocess.run([sys.executable, "-m", "deeploopcloser_amd.loop_closure", "--help"], capture_output=True, text=True,
                         cwd=ROOT, timeout=300, env=dict(os.environ, PYTHONPATH=ROOT))
    assert res.returncode == 0 and "usage:" in res.stdout and "dataset_path" in res.stdout
    res = subprocess.run([sys.executable, "-m", "deeploopcloser_amd.loop_closure"], capture_output=True, text=True,
                         cwd=ROOT, timeout=300, env=dict(os.environ, PYTHONPATH=ROOT))
    assert res.returncode == 2 and "dataset_path" in res.stderr            # argparse's "required" error, not silence
