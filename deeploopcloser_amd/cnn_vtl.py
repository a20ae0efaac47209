"""CnnVtl encoder with the reference's call surface, running on MI355X.

Mirrors src/cnn_vtl/network/cnn_vtl.py: ctor (:13-17), model (:28-128),
transform (:130-133).  Each convolution is one fp64 MFMA GEMM with the bias / ReLU fused, run as
an implicit GEMM (dlc_conv2d_nhwc_f64: the A-tile loader gathers input pixels straight from the
NHWC tensor, no im2col matrix -- 8 channels per load for conv2..conv5, element by element for
conv1's 3 input channels); pooling, per-row min/max, the 0..255 scaling, the int8 cast and the
column gather are HIP kernels too.

The reference's weights (bvlc_alexnet.npy, cnn_vtl.py:137-149) are a git-LFS
pointer upstream, so weights here are seeded synthetic AlexNet-shaped tensors
unless set_weights()/load_alexnet_npy() is used.  The reference's column mask
is drawn from the unseeded global NumPy RNG (:118-128); here the same draw
takes an explicit ``mask_seed``.
"""
import numpy as np
import torch

from . import _lib as L
from .engine import default_engine
from .math_utils import MathUtils

# (name, kh, kw, cin, cout, stride, padding, relu, pool_after) -- cnn_vtl.py:33-93
_LAYERS = (
    ("conv1", 11, 11, 3, 96, 4, "VALID", True, True),
    ("conv2", 5, 5, 96, 256, 1, "SAME", True, True),
    ("conv3", 3, 3, 256, 384, 1, "SAME", True, False),
    ("conv4", 3, 3, 384, 384, 1, "SAME", True, False),
    ("conv5", 3, 3, 384, 256, 1, "SAME", False, False),
)


def tf1_constant_fill(value, shape):
    """What ``tf.constant_initializer(value)`` (TF 1.x) yields for a variable of `shape`
    (cnn_vtl.py:137-149 hands it the arrays of bvlc_alexnet.npy as they are): the values in C
    order, and when there are FEWER than the shape holds the last one repeated to the end
    (tensor_util.make_tensor_proto + the Const kernel's fill); more values than the shape holds is
    TensorFlow's "Too many elements provided" ValueError.  The published blob stores AlexNet's
    grouped kernels -- conv2 (5,5,48,256), conv4 (3,3,192,384), conv5 (3,3,192,256) -- while the
    reference builds ungrouped 96->256 / 384->384 / 384->256 convolutions (:47-93), so half of each of
    those kernels is the blob in C order and the other half one repeated number."""
    v = np.asarray(value, dtype=np.float64).reshape(-1)
    need = int(np.prod(shape))
    if v.size > need:
        raise ValueError("Too many elements provided. Needed at most %d, but received %d" % (need, v.size))
    if v.size == 0:
        raise ValueError("constant_initializer needs at least one value")
    out = np.empty(need, dtype=np.float64)
    out[:v.size] = v
    out[v.size:] = v[-1]
    return out.reshape(shape)


def alexnet_params_from_dict(layer_params):
    """conv1..conv5 (HWIO kernels, biases) from the {layer: [W, b]} dict of bvlc_alexnet.npy the
    way the reference's graph ends up holding them (cnn_vtl.py:137-149: fc6-8 skipped, every array
    through tf.constant_initializer into the ungrouped variable shapes of :33-93).  Keys may be
    str or bytes (np.load(..., encoding='bytes'))."""
    def get(name):
        for key in (name, name.encode()):
            if key in layer_params:
                return layer_params[key]
        raise KeyError("layer %r missing from the weight dict" % name)
    ws, bs = [], []
    for name, kh, kw, cin, cout, *_ in _LAYERS:
        w, b = get(name)[0], get(name)[1]
        ws.append(tf1_constant_fill(w, (kh, kw, cin, cout)))
        bs.append(tf1_constant_fill(b, (cout,)))
    return ws, bs


def space_to_depth_kernel(w, s):
    """HWIO kernel [k, k, c, o] of a stride-s VALID convolution -> the [ceil(k/s), ceil(k/s), s*s*c, o] kernel of the
    equivalent stride-1 VALID convolution over the space-to-depth(s) input (include/dlc.h, dlc_space_to_depth_nhwc_f64):
    W'[ky', kx', (dy*s + dx)*c + ch] = W[ky'*s + dy, kx'*s + dx, ch], zero where an index reaches k."""
    kh, kw, c, o = w.shape
    k2h, k2w = -(-kh // s), -(-kw // s)
    wp = np.zeros((k2h * s, k2w * s, c, o), dtype=np.float64)
    wp[:kh, :kw] = w
    # [ky', dy, kx', dx, c, o] -> [ky', kx', dy, dx, c, o]
    return wp.reshape(k2h, s, k2w, s, c, o).transpose(0, 2, 1, 3, 4, 5).reshape(k2h, k2w, s * s * c, o)


def _out_size(n, k, s, padding):
    if padding == "VALID":
        return (n - k) // s + 1, 0
    o = -(-n // s)
    pad = max((o - 1) * s + k - n, 0)
    return o, pad // 2


class CnnVtl:
    def __init__(self, input_shape=(1, 224, 224, 3), batch_size: int = 10, compress_factor: float = 99.59,
                 seed=0, mask_seed=0, device=None, frame_chunk=2048):
        if len(input_shape) != 4 or any(int(v) <= 0 for v in input_shape) or input_shape[3] != 3:
            raise ValueError("input_shape must be [N, H, W, 3] with positive entries")
        if not (0 <= compress_factor <= 100):
            raise ValueError("compress_factor must be between 0 and 100")        # v8n rule, cnn_vtl.py:22
        self.input_shape = list(input_shape)
        self.batch_size = batch_size
        self.compress_factor = compress_factor
        self.frame_chunk = int(frame_chunk)
        self.engine = default_engine(device)
        # cnn_vtl.py:30,128 keep the graph's placeholder and output TENSORS as attributes (basic_example.py never reads
        # them; a caller could only sess.run them).  There is no graph here -- transform() is the eager equivalent -- so
        # the names exist and hold None: code that merely touches net.x / net.y keeps importing and running.
        self.x = None
        self.y = None
        self._define_model(seed, mask_seed)

    def _define_model(self, seed, mask_seed):
        h, w = self.input_shape[1], self.input_shape[2]
        self._geom, self.layer_sizes = [], []
        for name, kh, kw, cin, cout, s, pad, relu, pool in _LAYERS:
            oh, ph = _out_size(h, kh, s, pad)
            ow, pw = _out_size(w, kw, s, pad)
            self._geom.append((kh, kw, cin, cout, s, ph, pw, oh, ow, relu, pool))
            self.layer_sizes.append(oh * ow * cout)
            h, w = oh, ow
            if pool:
                h, w = (h - 3) // 2 + 1, (w - 3) // 2 + 1
        rng = np.random.RandomState(seed)
        ws, bs = [], []
        for _, kh, kw, cin, cout, _, _, _, _ in _LAYERS:
            ws.append(rng.standard_normal((kh, kw, cin, cout)) / np.sqrt(kh * kw * cin))
            bs.append(rng.standard_normal(cout) * 0.1)
        self.set_weights(ws, bs)
        # column mask, cnn_vtl.py:118-128 (choice WITH replacement, union)
        mrng = np.random.RandomState(mask_seed)
        mask = np.zeros(int(np.sum(self.layer_sizes)), dtype=bool)
        start = 0
        for size in self.layer_sizes:
            idx = mrng.choice(np.arange(start, start + size), size=MathUtils.compressed_size(size, self.compress_factor))
            start += size
            mask[idx] = True
        self.set_columns(np.nonzero(mask)[0])

    def set_columns(self, columns):
        columns = np.asarray(columns, dtype=np.int64)
        total = int(np.sum(self.layer_sizes))
        if columns.ndim != 1 or columns.size == 0 or columns.min() < 0 or columns.max() >= total:
            raise ValueError("columns must be a non-empty 1-D index list into the %d-wide descriptor" % total)
        self.columns = columns
        self._columns_dev = torch.from_numpy(columns).to(self.engine.device)

    def set_weights(self, weights, biases):
        """HWIO kernels + biases for conv1..conv5."""
        ws, bs = [], []
        for (name, kh, kw, cin, cout, *_), w, b in zip(_LAYERS, weights, biases):
            w = np.asarray(w, dtype=np.float64)
            if w.shape != (kh, kw, cin, cout):
                raise ValueError("%s kernel must be %s, got %s" % (name, (kh, kw, cin, cout), w.shape))
            ws.append(self.engine.to_device(w.reshape(kh * kw * cin, cout), torch.float64))
            bs.append(self.engine.to_device(np.asarray(b, dtype=np.float64).reshape(cout), torch.float64))
        self._w, self._b = ws, bs
        # Strided VALID layers whose input is a whole number of stride blocks (conv1: 11x11 / 4 on 192x240 or 224x224)
        # run as a stride-1 convolution over the space-to-depth input: 3 input channels become 48, which the implicit
        # GEMM's loaders fetch 8 / 16 at a time (the element-wise gather of a 3-channel input ran at 30 TF, the
        # channel-tiled form at 50-60).  Same products; the zero taps of the padded 12x12 kernel add nothing.
        self._s2d = {}
        for l, ((name, kh, kw, cin, cout, s, pad, *_), w) in enumerate(zip(_LAYERS, weights)):
            if l == 0 and s > 1 and pad == "VALID" and self.input_shape[1] % s == 0 and self.input_shape[2] % s == 0 \
                    and (s * s * cin) % 8 == 0:
                wp = space_to_depth_kernel(np.asarray(w, dtype=np.float64), s)
                self._s2d[l] = (s, wp.shape[0], wp.shape[1],
                                self.engine.to_device(wp.reshape(-1, cout), torch.float64))

    def load_alexnet_npy(self, path):
        """The {layer: [W, b]} dict layout of bvlc_alexnet.npy (cnn_vtl.py:137-149), fc6-8 skipped;
        grouped kernels are filled the way TF-1's constant_initializer fills them (tf1_constant_fill)."""
        d = np.load(path, encoding="bytes", allow_pickle=True).item()
        self.set_weights(*alexnet_params_from_dict(d))

    def _features(self, x, frame_keys=None):
        """conv1..conv5 outputs of a frame chunk: list of [n, oh, ow, cout] fp64 tensors.  frame_keys: every layer
        folds the per-frame minimum / maximum of its outputs into it (the descriptor's range, cnn_vtl.py:110-112)."""
        e = self.engine
        outs = []
        h = x
        n = x.shape[0]
        for l, ((kh, kw, cin, cout, s, ph, pw, oh, ow, relu, pool), w, b) in enumerate(zip(self._geom, self._w, self._b)):
            act = L.DLC_ACT_RELU if relu else L.DLC_ACT_NONE
            if l in self._s2d:
                bs_, k2h, k2w, wp = self._s2d[l]
                y = e.conv2d(e.space_to_depth(h, bs_), wp, b, k2h, k2w, 1, 0, 0, oh, ow, act, frame_keys=frame_keys)
            else:
                y = e.conv2d(h, w, b, kh, kw, s, ph, pw, oh, ow, act, frame_keys=frame_keys)
            outs.append(y)
            h = e.maxpool3x3s2(y) if pool else y
        return outs

    def transform_tensor(self, x):
        x = self.engine.to_device(x, torch.float64)
        if x.dim() != 4 or list(x.shape[1:]) != list(self.input_shape[1:]):
            raise ValueError("expected input of shape [N, %d, %d, 3], got %s" %
                             (self.input_shape[1], self.input_shape[2], tuple(x.shape)))
        # frame chunks of at most frame_chunk frames (activations: 6.3 MB per 192x240 frame), EQUAL to within one
        # frame: 1063 frames cut 504 + 504 + 55 spent 10 % of the call on the last 5 % of the frames, whose launches
        # are too small for the large-tile kernels.  A frame's descriptor does not depend on its chunk (tests).
        parts = []
        n = x.shape[0]
        n_chunks = max(1, -(-n // max(1, self.frame_chunk)))
        step = -(-n // n_chunks) if n else 1
        for lo in range(0, n, step):
            chunk = x[lo:lo + step].contiguous()
            if chunk.shape[0] <= 65535:
                keys = self.engine.frame_minmax_keys(chunk.shape[0])
                parts.append(self.engine.quant_gather(self._features(chunk, keys), self._columns_dev, keys))
            else:
                parts.append(self.engine.minmax_quant_gather(self._features(chunk), self._columns_dev))
        if not parts:
            return torch.empty((0, self.columns.size), dtype=torch.int8, device=self.engine.device)
        return torch.cat(parts, dim=0)

    def transform(self, x, chunk_frames=None):
        """CnnVtl.transform (cnn_vtl.py:130-133): frames [N,H,W,3] -> int8 [N, D'].
        A host array (uint8 frames as cv2.imread returns them, or float64 as the reference's placeholder holds them) is
        encoded in equal chunks of at most `chunk_frames` frames (default: a quarter of frame_chunk) whose upload and
        convolutions overlap (Engine.run_chunked); a frame's descriptor does not depend on its chunk."""
        if isinstance(x, torch.Tensor):
            return self.engine.download(self.transform_tensor(x))
        x = np.asarray(x)
        if x.ndim != 4 or list(x.shape[1:]) != list(self.input_shape[1:]):
            raise ValueError("expected input of shape [N, %d, %d, 3], got %s" %
                             (self.input_shape[1], self.input_shape[2], tuple(x.shape)))
        n = x.shape[0]
        if n == 0:
            return np.empty((0, self.columns.size), dtype=np.int8)
        if x.dtype not in (np.uint8, np.float64, np.float32):
            x = x.astype(np.float64)
        cf = int(chunk_frames or max(1, self.frame_chunk // 4))
        n_chunks = max(1, -(-n // cf))
        step = -(-n // n_chunks)
        return self.engine.run_chunked(x, step, self.transform_tensor)
