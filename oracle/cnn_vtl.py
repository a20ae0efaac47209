"""NumPy fp64 restatement of the CnnVtl encoder
(src/cnn_vtl/network/cnn_vtl.py:28-133).  Test infrastructure only.
PARITY UNPINNED by the reference (TensorFlow not installable here; the AlexNet
weight blob is a git-LFS pointer, so weights are seeded synthetic ones).

Defined behaviour where the reference is implementation-defined:
  * ``tf.cast(float64 -> int8)`` of values in [0,255] (cnn_vtl.py:115-116):
    truncation toward zero, then wrap modulo 256 (128..255 -> -128..-1), which
    is what the x86 Eigen scalar path does.
  * the column mask (cnn_vtl.py:118-128) uses the UNSEEDED global NumPy RNG in
    the reference; here it takes an explicit seed, same draw sequence
    (``choice`` with replacement per layer, union of the draws)."""
import numpy as np

from .math_utils import compressed_size

# (name, kh, kw, cin, cout, stride, padding, relu) -- cnn_vtl.py:33-93
LAYERS = (
    ("conv1", 11, 11, 3, 96, 4, "VALID", True),
    ("conv2", 5, 5, 96, 256, 1, "SAME", True),
    ("conv3", 3, 3, 256, 384, 1, "SAME", True),
    ("conv4", 3, 3, 384, 384, 1, "SAME", True),
    ("conv5", 3, 3, 384, 256, 1, "SAME", False),
)
POOL_AFTER = ("conv1", "conv2")       # 3x3 stride 2 VALID, cnn_vtl.py:42-45,58-61


def init_weights(seed, scale="fan_in"):
    """Seeded AlexNet-conv-shaped HWIO weights + biases (the reference reads
    them from bvlc_alexnet.npy, cnn_vtl.py:137-149, which is absent)."""
    rng = np.random.RandomState(seed)
    ws, bs = [], []
    for _, kh, kw, cin, cout, _, _, _ in LAYERS:
        w = rng.standard_normal((kh, kw, cin, cout))
        if scale == "fan_in":
            w = w / np.sqrt(kh * kw * cin)
        ws.append(w)
        bs.append(rng.standard_normal(cout) * 0.1)
    return ws, bs


def constant_initializer_fill(value, shape):
    """tf.constant_initializer(value) into a variable of `shape` (cnn_vtl.py:137-149, TF 1.x):
    C-order values, the last one repeated when there are fewer than the shape holds; more is an
    error.  This is how bvlc_alexnet.npy's grouped kernels (conv2 [5,5,48,256], conv4 [3,3,192,384],
    conv5 [3,3,192,256]) land in the reference's ungrouped conv variables (:47-93)."""
    flat = np.asarray(value, dtype=np.float64).ravel()
    total = int(np.prod(shape))
    if flat.size > total:
        raise ValueError("Too many elements provided")
    return np.concatenate([flat, np.full(total - flat.size, flat[-1])]).reshape(shape)


def weights_from_alexnet_dict(layer_params):
    """(weights, biases) of conv1..conv5 from the {layer: [W, b]} dict (fc6-8 skipped, :141-148)."""
    ws, bs = [], []
    for name, kh, kw, cin, cout, _, _, _ in LAYERS:
        entry = layer_params[name] if name in layer_params else layer_params[name.encode()]
        ws.append(constant_initializer_fill(entry[0], (kh, kw, cin, cout)))
        bs.append(constant_initializer_fill(entry[1], (cout,)))
    return ws, bs


def _out_size(n, k, s, padding):
    if padding == "VALID":
        return (n - k) // s + 1, 0
    o = -(-n // s)                                   # ceil
    pad = max((o - 1) * s + k - n, 0)
    return o, pad // 2                               # TF SAME: extra pad goes after


def conv2d_nhwc(x, w, b, stride, padding, relu):
    """tf.layers.conv2d (NHWC input, HWIO kernel) as im2col + one fp64 GEMM.
    im2col column order is (kh, kw, cin), matching w.reshape(kh*kw*cin, cout)."""
    n, h, wd, c = x.shape
    kh, kw, cin, cout = w.shape
    assert cin == c
    oh, ph = _out_size(h, kh, stride, padding)
    ow, pw = _out_size(wd, kw, stride, padding)
    if padding == "SAME":
        hp = max((oh - 1) * stride + kh, ph + h)     # stride > 1 can leave trailing pixels unused (no pad needed)
        wp = max((ow - 1) * stride + kw, pw + wd)
        xp = np.zeros((n, hp, wp, c), dtype=np.float64)
        xp[:, ph:ph + h, pw:pw + wd, :] = x
    else:
        xp = x
    cols = np.empty((n, oh, ow, kh, kw, c), dtype=np.float64)
    for i in range(kh):
        for j in range(kw):
            cols[:, :, :, i, j, :] = xp[:, i:i + (oh - 1) * stride + 1:stride,
                                        j:j + (ow - 1) * stride + 1:stride, :]
    y = cols.reshape(n * oh * ow, kh * kw * c) @ w.reshape(kh * kw * c, cout) + b
    if relu:
        y = np.maximum(y, 0.0)
    return y.reshape(n, oh, ow, cout)


def maxpool3x3s2(x):
    """tf.layers.max_pooling2d(pool 3x3, stride 2, VALID) (cnn_vtl.py:42-45)."""
    n, h, w, c = x.shape
    oh, ow = (h - 3) // 2 + 1, (w - 3) // 2 + 1
    out = np.full((n, oh, ow, c), -np.inf)
    for i in range(3):
        for j in range(3):
            out = np.maximum(out, x[:, i:i + 2 * (oh - 1) + 1:2, j:j + 2 * (ow - 1) + 1:2, :])
    return out


def layer_sizes(input_hw):
    """Flattened per-frame sizes of conv1..conv5 outputs (cnn_vtl.py:98)."""
    h, w = input_hw
    sizes = []
    for name, kh, kw, _, cout, s, pad, _ in LAYERS:
        h, _ = _out_size(h, kh, s, pad)
        w, _ = _out_size(w, kw, s, pad)
        sizes.append(h * w * cout)
        if name in POOL_AFTER:
            h, w = (h - 3) // 2 + 1, (w - 3) // 2 + 1
    return sizes


def column_indices(sizes, compress_factor=99.59, seed=0):
    """The boolean column mask of cnn_vtl.py:118-128 as a sorted index list.
    Per layer ``np.random.choice(arange(start, start+size), size=compressed_size)``
    (with replacement -> duplicates collapse, so the list is <= the sum)."""
    rng = np.random.RandomState(seed)
    mask = np.zeros(int(np.sum(sizes)), dtype=bool)
    start = 0
    for i in sizes:
        idx = rng.choice(np.arange(start, start + i), size=compressed_size(i, compress_factor))
        start += i
        mask[idx] = True
    return np.nonzero(mask)[0].astype(np.int64)


def features(x, weights, biases):
    """conv1..conv5 outputs flattened (NHWC order) and concatenated:
    d [N, sum(sizes)] fp64 (cnn_vtl.py:33-106)."""
    x = np.asarray(x, dtype=np.float64)
    n = x.shape[0]
    outs = []
    h = x
    for (name, _, _, _, _, s, pad, relu), w, b in zip(LAYERS, weights, biases):
        h = conv2d_nhwc(h, w, b, s, pad, relu)
        outs.append(h.reshape(n, -1))
        if name in POOL_AFTER:
            h = maxpool3x3s2(h)
    return np.concatenate(outs, axis=1)


def quantize_int8(d):
    """cnn_vtl.py:108-116: per-row min/max, (d-min)*(255/(max-min)), cast int8
    (truncate toward zero, wrap mod 256)."""
    mx = d.max(axis=1).reshape(-1, 1)
    mn = d.min(axis=1).reshape(-1, 1)
    scaled = (d - mn) * (np.float64(255) / (mx - mn))
    t = np.trunc(scaled).astype(np.int64)
    return (t & 0xFF).astype(np.uint8).view(np.int8)


def transform(x, weights, biases, columns):
    """CnnVtl.transform (cnn_vtl.py:130-133): int8 [N, len(columns)]."""
    return quantize_int8(features(x, weights, biases))[:, columns]
