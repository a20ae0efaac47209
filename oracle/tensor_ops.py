"""NumPy fp64 restatement of the TensorWrapper op vocabulary the SDAV forward
is written in (src/utils/TensorflowWrapper.py).  Test infrastructure only."""
import numpy as np


def flat_batch(x):
    """TensorWrapper.flat_batch (TensorflowWrapper.py:13-15): [B,P,K] -> [B*P,K]."""
    s = x.shape
    return x.reshape(s[0] * s[1], s[2])


def batch(x, batch_size):
    """TensorWrapper.batch (TensorflowWrapper.py:17-20): [B*P,N] -> [B,P,N]."""
    s = x.shape
    return x.reshape(batch_size, s[0] // batch_size, s[1])


def tw_matmul(x, y):
    """TensorWrapper.matmul (TensorflowWrapper.py:57-67).

    Equal ranks: plain ``x @ y``.  Otherwise (3-D x, 2-D y): flatten the batch,
    multiply, re-batch.  Pinned by test/TensorflowWrapperTest.py:11-21.
    """
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    if x.ndim == y.ndim:
        return x @ y
    b = x.shape[0] if x.ndim == 3 else 1           # batch_size(), :22-26
    return batch(flat_batch(x) @ y, b)


def sigmoid(z):
    """TensorWrapper.sigmoid -> tf.nn.sigmoid (TensorflowWrapper.py:77-78):
    logistic 1/(1+exp(-z)) in fp64."""
    with np.errstate(over="ignore"):
        return 1.0 / (1.0 + np.exp(-z))


def corruption_mask(shape, level, rng):
    """random_mask (TensorflowWrapper.py:148-156): round(P*K*level) zeros
    (tf.round = half-to-even), the rest ones, shuffled, shape [P,K]; shared by
    the whole batch (corrupt(), :34-38).  At transform level == 0 -> all ones."""
    n = int(shape[0]) * int(shape[1])
    n_zeros = int(np.round(n * float(level)))      # np.round is half-to-even too
    m = np.concatenate([np.ones(n - n_zeros), np.zeros(n_zeros)])
    rng.shuffle(m)
    return m.reshape(shape)
