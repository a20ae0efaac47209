"""Eager MI355X counterpart of the reference's tensor-op facade
(src/utils/TensorflowWrapper.py:6-156): the same method names, on torch-ROCm
tensors, with matmul / add / sigmoid running in the HIP kernels.

``x.matmul(w).add(b).sigmoid()`` is fused: matmul returns a wrapper with a
pending GEMM, add() attaches the bias, sigmoid() the activation, and the chain
is materialised as ONE dlc_gemm_bias_act call when its value is needed.
"""
import numpy as np
import torch

from . import _lib as L
from .engine import default_engine


def parameter_guard(y):                             # TensorflowWrapper.py:117-123
    if isinstance(y, TensorWrapper):
        return y.to_tf()
    if isinstance(y, list):
        return list(map(parameter_guard, y))
    return y


def _as_tensor(v, dtype=torch.float64):
    e = default_engine()
    if isinstance(v, torch.Tensor):
        return v.to(e.device)
    return e.to_device(np.asarray(v), dtype)


class TensorWrapper:
    def __init__(self, x, _pending=None):
        if isinstance(x, TensorWrapper):
            x = x.to_tf()
        self._x = x
        self._pending = _pending          # (a2d, w, bias, act, out_shape)

    # ---- materialisation -----------------------------------------------------------
    @property
    def x(self):
        return self.to_tf()

    def to_tf(self):                                # :113-114 (a torch tensor here)
        if self._pending is not None:
            a, w, bias, act, shape = self._pending
            self._x = default_engine().gemm_bias_act(a, w, bias, act=act).reshape(shape)
            self._pending = None
        return self._x

    def numpy(self):
        return self.to_tf().cpu().numpy()

    # ---- shape ops (views) ------------------------------------------------------------
    def dimensions(self):                           # :43-44
        return len(self._shape())

    def _shape(self):
        return tuple(self._pending[4]) if self._pending is not None else tuple(self._x.shape)

    def shape(self):                                # :46-47
        return list(self._shape())

    def rank(self):                                 # :40-41
        return self.dimensions()

    def batch_size(self):                           # :22-26
        return self._shape()[0] if self.dimensions() == 3 else 1

    def parameter_number(self):                     # :28-32
        s = self._shape()
        return s[1] * s[2] if self.dimensions() == 3 else s[0] * s[1]

    def reshape(self, shape):                       # :53-55
        return TensorWrapper(self.to_tf().reshape([int(v) for v in parameter_guard(shape)]))

    def flat_batch(self):                           # :13-15
        s = self._shape()
        return self.reshape([s[0] * s[1], s[2]])

    def batch(self, batch_size):                    # :17-20
        s = self._shape()
        return self.reshape([int(batch_size), s[0] // int(batch_size), s[1]])

    def concat(self, y, axis=0):                    # :49-51
        return TensorWrapper(torch.cat([self.to_tf(), _as_tensor(parameter_guard(y), self.to_tf().dtype)], dim=axis))

    def to(self, dtype):                            # :86-87
        return TensorWrapper(self.to_tf().to(dtype))

    # ---- arithmetic -----------------------------------------------------------------------
    def matmul(self, y):                            # :57-67
        y = _as_tensor(parameter_guard(y), torch.float64)
        x = self.to_tf()
        if x.dtype != y.dtype:
            y = y.to(x.dtype)
        if x.dim() == y.dim() == 2:
            return TensorWrapper(None, _pending=(x, y, None, L.DLC_ACT_NONE, (x.shape[0], y.shape[1])))
        if x.dim() == 3 and y.dim() == 2:          # broadcast: flatten, multiply, re-batch
            b = x.shape[0]
            x2 = x.reshape(x.shape[0] * x.shape[1], x.shape[2])
            return TensorWrapper(None, _pending=(x2, y, None, L.DLC_ACT_NONE, (b, x2.shape[0] // b, y.shape[1])))
        raise ValueError("matmul: unsupported ranks %d x %d" % (x.dim(), y.dim()))

    def add(self, y):                               # :69-71
        y = _as_tensor(parameter_guard(y), torch.float64)
        if self._pending is not None and self._pending[2] is None and self._pending[3] == L.DLC_ACT_NONE \
                and y.dim() == 1 and y.numel() == self._pending[4][-1]:
            a, w, _, act, shape = self._pending
            return TensorWrapper(None, _pending=(a, w, y.to(a.dtype).contiguous(), act, shape))
        x = self.to_tf()
        if y.dim() == 1 and y.numel() == x.shape[-1]:
            return TensorWrapper(default_engine().bias_act(x, y, L.DLC_ACT_NONE))
        return TensorWrapper(x + y.to(x.dtype))     # general broadcast: plumbing only, not on the hot path

    def sigmoid(self):                              # :77-78
        if self._pending is not None and self._pending[3] == L.DLC_ACT_NONE:
            a, w, bias, _, shape = self._pending
            return TensorWrapper(None, _pending=(a, w, bias, L.DLC_ACT_SIGMOID, shape))
        return TensorWrapper(default_engine().bias_act(self.to_tf(), None, L.DLC_ACT_SIGMOID))

    def multiply(self, y):                          # :73-75
        return TensorWrapper(self.to_tf() * _as_tensor(parameter_guard(y), self.to_tf().dtype))

    def corrupt(self, corruption_level, generator=None):   # :34-38
        s = self._shape()
        s = s[1:] if self.dimensions() == 3 else s
        return self.multiply(random_mask(list(s), corruption_level, generator=generator))

    def round(self):                                # :83-84 (tf.round = half to even, as torch.round)
        return TensorWrapper(torch.round(self.to_tf()))

    def shuffle(self, generator=None):              # :80-81 (shuffles along axis 0)
        x = self.to_tf()
        perm = torch.randperm(x.shape[0], device=x.device, generator=generator)
        return TensorWrapper(x[perm])

    def __truediv__(self, y):
        return TensorWrapper(self.to_tf() / parameter_guard(y))

    def __getitem__(self, item):
        return TensorWrapper(self.to_tf()[parameter_guard(item)])

    def __mul__(self, y):
        return self.multiply(y)

    __rmul__ = __mul__

    def __add__(self, y):
        return self.add(y)

    def __sub__(self, y):
        return TensorWrapper(self.to_tf() - _as_tensor(parameter_guard(y), self.to_tf().dtype))


def constant(value, shape=None, dtype=torch.float64):          # :130-135
    t = _as_tensor(np.asarray(value, dtype=np.float64), dtype).to(dtype)
    if shape:
        t = t.expand([int(v) for v in parameter_guard(shape)]).contiguous() if t.dim() == 0 else t.reshape(shape)
    return TensorWrapper(t)


def placeholder(dtype, shape):                                  # :126-127 (eager: zeros of that shape)
    return zeros([0 if v is None else v for v in shape], dtype=dtype)


def zeros(shape, dtype=torch.float64):                          # :138-140
    return TensorWrapper(torch.zeros([int(v) for v in parameter_guard(shape)], dtype=dtype, device=default_engine().device))


def ones(shape, dtype=torch.float64):                           # :143-145
    return TensorWrapper(torch.ones([int(v) for v in parameter_guard(shape)], dtype=dtype, device=default_engine().device))


def random_mask(shape, zeros_percentage, dtype=torch.float64, generator=None):   # :148-156
    shape = [int(v) for v in parameter_guard(shape)]
    parameters = shape[0] * shape[1]
    n_zeros = int(np.round(parameters * float(zeros_percentage)))
    n_ones = parameters - n_zeros
    return ones([n_ones], dtype=dtype).concat(zeros([n_zeros], dtype=dtype)).shuffle(generator).reshape(shape)
