"""The encoder oracles are "parity unpinned" against the reference (TensorFlow is not installable
here).  These CPU tests pin them against INDEPENDENT implementations of the same operators that are
available -- PyTorch's CPU kernels and SciPy -- so that an error in the restatement would have to be
made twice, in two unrelated code bases, to go unnoticed.  TensorFlow's padding convention for
'SAME' (extra pad after) is stated explicitly in the comparison."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import cnn_vtl as ocnn
from oracle import sdav as osdav


def tf_same_pads(n, k, s):
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return total // 2, total - total // 2            # (before, after): TensorFlow puts the odd pixel after


@pytest.mark.parametrize("h,w,kh,kw,cin,cout,stride,padding,relu", [
    (27, 31, 11, 11, 3, 8, 4, "VALID", True),        # conv1's geometry (11x11 stride 4 VALID)
    (22, 28, 5, 5, 6, 10, 1, "SAME", True),          # conv2 (5x5 SAME)
    (10, 13, 3, 3, 16, 12, 1, "SAME", False),        # conv5 (3x3 SAME, no activation)
    (12, 9, 3, 3, 4, 5, 2, "SAME", True),            # stride-2 SAME: asymmetric pads (not in the model, pins the rule)
    (8, 6, 1, 1, 4, 3, 2, "SAME", True),             # 1x1 stride 2: no padding and a trailing row / column left unused
])
def test_conv2d_oracle_equals_torch_and_scipy(h, w, kh, kw, cin, cout, stride, padding, relu):
    from scipy.signal import correlate
    rng = np.random.RandomState(h * w)
    x = rng.standard_normal((2, h, w, cin))
    k = rng.standard_normal((kh, kw, cin, cout))
    b = rng.standard_normal(cout)
    got = ocnn.conv2d_nhwc(x, k, b, stride, padding, relu)
    xt = torch.from_numpy(x).permute(0, 3, 1, 2)                               # NHWC -> NCHW
    kt = torch.from_numpy(k).permute(3, 2, 0, 1)                               # HWIO -> OIHW
    if padding == "SAME":
        (pt, pb), (pl, pr) = tf_same_pads(h, kh, stride), tf_same_pads(w, kw, stride)
        xt = F.pad(xt, (pl, pr, pt, pb))
    ref = F.conv2d(xt, kt, torch.from_numpy(b), stride=stride).permute(0, 2, 3, 1).numpy()
    if relu:
        ref = np.maximum(ref, 0.0)
    assert got.shape == ref.shape and np.abs(got - ref).max() < 1e-12
    # SciPy: full correlation of the padded image, one output channel, sampled at the stride
    xp = xt.permute(0, 2, 3, 1).numpy()[0]
    full = sum(correlate(xp[:, :, c], k[:, :, c, 0], mode="valid") for c in range(cin)) + b[0]
    full = full[::stride, ::stride]
    if relu:
        full = np.maximum(full, 0.0)
    assert np.abs(got[0, :, :, 0] - full).max() < 1e-12


def test_maxpool_and_quantise_oracle_equal_torch():
    rng = np.random.RandomState(4)
    x = rng.standard_normal((2, 46, 58, 5))
    got = ocnn.maxpool3x3s2(x)
    ref = F.max_pool2d(torch.from_numpy(x).permute(0, 3, 1, 2), 3, 2).permute(0, 2, 3, 1).numpy()
    assert np.array_equal(got, ref)
    d = rng.standard_normal((3, 1000)) * 50
    q = ocnn.quantize_int8(d)
    # plain Python floats (IEEE double, one operation at a time), as cnn_vtl.py:108-116 writes it:
    # (d - min) * (255 / (max - min)), truncate toward zero, wrap mod 256 into int8
    ref = np.empty(d.shape, dtype=np.int8)
    for r, row in enumerate(d.tolist()):
        mn, mx = min(row), max(row)
        scale = 255.0 / (mx - mn)
        for c, v in enumerate(row):
            t = int((v - mn) * scale) & 0xFF
            ref[r, c] = t - 256 if t > 127 else t
    assert np.array_equal(q, ref) and (q < 0).any() and q.max() == 127


def test_sdav_oracle_equals_torch():
    rng = np.random.RandomState(9)
    dims = [1681, 64, 48, 32]
    ws = [rng.standard_normal((dims[i], dims[i + 1])) for i in range(3)]
    bs = [rng.standard_normal(dims[i + 1]) for i in range(3)]
    x = rng.uniform(0, 1, (4, 30, 1681))
    got = osdav.transform(x, ws, bs)
    h = torch.from_numpy(x).reshape(120, 1681)
    for w, b in zip(ws, bs):
        h = torch.sigmoid(h @ torch.from_numpy(w) + torch.from_numpy(b))
    assert got.shape == (120, 32) and np.abs(got - h.numpy()).max() < 1e-12
