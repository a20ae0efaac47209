"""NumPy fp64 restatement of the SDAV / DA encoder forward.

Follows src/sdav/network/SDAV.py:120-163 (graph), :188-217 (variables),
:293-302 (transform) and src/sdav/network/DenoisingAutoencoderVariant.py:92-101,
116-119, 254-259.  Test infrastructure only.  PARITY UNPINNED by the reference
(no TensorFlow here, no golden vectors or weights upstream)."""
import numpy as np

from .tensor_ops import tw_matmul, sigmoid, flat_batch

INPUT_SHAPE = (30, 1681)            # SDAV.py:31
HIDDEN_UNITS = (2500,) * 5          # SDAV.py:32


def init_weights(seed, input_dim=INPUT_SHAPE[1], hidden_units=HIDDEN_UNITS, scale="reference"):
    """SDAV._define_model_variables (SDAV.py:188-217): W_l ~ N(0,1) fp64
    ([in,out]), b_l = 0.  ``scale='fan_in'`` divides by sqrt(fan_in) (a sanely
    scaled regime the reference does not have; used to stress numerics)."""
    rng = np.random.RandomState(seed)
    dims = [input_dim] + list(hidden_units)
    ws, bs = [], []
    for k, n in zip(dims[:-1], dims[1:]):
        w = rng.standard_normal((k, n))
        if scale == "fan_in":
            w = w / np.sqrt(k)
        ws.append(w)
        bs.append(np.zeros(n))
    return ws, bs


def transform(x, weights, biases):
    """SDAV.transform (SDAV.py:293-302) with corruption level 0.

    x: [B,30,K0] fp64.  Layer 0 is a 3-D x 2-D broadcast matmul (SDAV.py:129),
    layers 1..4 work on the flattened [B*30, H] batch (:134-157); the corruption
    mask is all ones at level 0 (TensorflowWrapper.py:148-156) so it is the
    identity.  Returns h4 as the FLAT [B*30, H] array (SDAV.py:163,302)."""
    x = np.asarray(x, dtype=np.float64)
    h = sigmoid(tw_matmul(x, weights[0]) + biases[0])          # [B,30,H]
    h = flat_batch(h)
    for w, b in zip(weights[1:], biases[1:]):
        h = sigmoid(h @ w + b)
    return h


def da_transform(x, w, b):
    """DA.transform (DenoisingAutoencoderVariant.py:116-119,254-259):
    sigmoid(x @ W + b) on one frame [30,in] -> [30,H]."""
    return sigmoid(np.asarray(x, dtype=np.float64) @ w + b)
