"""NumPy fp64 restatement of one SDAV training step (test infrastructure only).

Follows src/sdav/network/SDAV.py: forward graph :126-159, per-layer loss :171-186,
plain SGD :223-226 (``optimizer.minimize(loss_l)`` differentiates with respect to EVERY
trainable variable loss_l depends on, so training layer l also moves the encoders of
layers 0..l-1), tied decoders :192-216, masking noise TensorflowWrapper.py:34-38,148-156.
PARITY UNPINNED by the reference (TensorFlow not installable); the analytic gradients below
are pinned by finite differences in tests/test_train_oracle.py.

Quirks of the reference that ARE the spec and are reproduced:
  * the "cross entropy" is softmax_cross_entropy_with_logits_v2(labels=x, logits=y) with
    y = sigmoid(...) used as LOGITS and x (not a distribution) as labels (:172), and the v2 op
    back-propagates into the labels too;
  * for layer 0 the labels are the UNcorrupted input, for layers >= 1 the corrupted one
    (x_l is defined after .corrupt(), :135,142,149,156);
  * the mask is [P, K], shared by all frames of the batch;
  * cc slices hidden_units[0] columns of `frames` and hidden_units[l] of `frames_next` (:178-181;
    equal widths in the shipped configuration, required equal here);
  * the sparsity term cs = reduce_mean(tf.norm(h - s, axis=1, ord=1)) (:174) sees a 3-D h
    [B, P, N] at layer 0 (the 3-D x 2-D matmul of TensorflowWrapper.py:57-67 re-batches its result,
    SDAV.py:129-131) and a 2-D h [B*P, N] at layers >= 1 (flat_batch, :135-157): axis 1 is the
    PATCH axis at layer 0 (sum over P, mean over B*N entries) and the UNIT axis afterwards (sum
    over N, mean over B*P rows).  Same numerator, denominators B*N vs B*P: layer 0's cs and its
    gradient are P/N (30/2500) of what the 2-D form would give.
A batch of one frame makes cc the mean of an empty tensor (NaN in TensorFlow); this
restatement, like the MI355X build, requires batch >= 2.
"""
import numpy as np

from .tensor_ops import sigmoid


def forward_layer(x_in, mask, w, b_enc, b_dec):
    """x_in [B,P,K] -> (x_tilde [R,K], h [R,N], y [R,K]) with R = B*P."""
    b, p, k = x_in.shape
    xt = (x_in * mask[None, :, :]).reshape(b * p, k)
    h = sigmoid(xt @ w + b_enc)
    y = sigmoid(h @ w.T + b_dec)
    return xt, h, y


def sparsity_denominator(layer, batch, patches, units):
    """Number of entries reduce_mean averages over in cs (SDAV.py:174): h is [B,P,N] at layer 0
    (norm over axis 1 = patches -> [B,N]) and [B*P,N] afterwards (norm over axis 1 = units -> [B*P])."""
    return batch * units if layer == 0 else batch * patches


def layer_loss(labels, h, y, batch, patches, sparse_level=0.05, sparse_penalty=1.0, consecutive_penalty=0.2, layer=1):
    """_define_loss_for_layer (SDAV.py:171-186): (loss, cd, cs, cc).  `layer` selects the shape h has
    in the reference's graph (3-D at layer 0, 2-D afterwards), which changes the cs normalisation."""
    ymax = y.max(axis=1, keepdims=True)
    logsm = y - ymax - np.log(np.exp(y - ymax).sum(axis=1, keepdims=True))
    cd = np.mean(-(labels * logsm).sum(axis=1))
    hb = h.reshape(batch, patches, -1)
    if layer == 0:
        cs = np.mean(np.abs(hb - sparse_level).sum(axis=1))       # [B,N]: summed over the patch axis
    else:
        cs = np.mean(np.abs(h - sparse_level).sum(axis=1))        # [B*P]: summed over the unit axis
    diff = hb[:-1] - hb[1:]
    cc = np.mean(np.sqrt((diff ** 2).sum(axis=(1, 2))))
    return cd + sparse_penalty * cs + consecutive_penalty * cc, cd, cs, cc


def loss_and_grads(layer, x, masks, ws, b_encs, b_dec, sparse_level=0.05, sparse_penalty=1.0,
                   consecutive_penalty=0.2):
    """Loss of `layer` and its gradients w.r.t. W_0..W_layer, b_enc_0..b_enc_layer and
    b_dec of `layer`.  x [B,P,K0]; masks[l] [P, dims[l]]; ws[l] [dims[l], dims[l+1]]."""
    batch, patches, _ = x.shape
    if batch < 2:
        raise ValueError("a training batch needs at least 2 frames (the consecutive-frame term)")
    rows = batch * patches
    # ---- forward through layers 0..layer, keeping what the backward pass needs
    xts, hs = [], []
    cur = x
    for l in range(layer + 1):
        xt = (cur * masks[l][None]).reshape(rows, -1)
        h = sigmoid(xt @ ws[l] + b_encs[l])
        xts.append(xt)
        hs.append(h)
        cur = h.reshape(batch, patches, -1)
    w, xt, h = ws[layer], xts[layer], hs[layer]
    y = sigmoid(h @ w.T + b_dec)
    labels = x.reshape(rows, -1) if layer == 0 else xt            # :131 vs :138-159
    loss, cd, cs, cc = layer_loss(labels, h, y, batch, patches, sparse_level, sparse_penalty, consecutive_penalty,
                                  layer=layer)

    # ---- backward
    ymax = y.max(axis=1, keepdims=True)
    e = np.exp(y - ymax)
    sm = e / e.sum(axis=1, keepdims=True)
    logsm = np.log(sm)
    d_y = (sm * labels.sum(axis=1, keepdims=True) - labels) / rows
    d_labels = -logsm / rows                                      # v2 back-propagates into the labels
    d_z2 = d_y * y * (1 - y)
    g_w = d_z2.T @ h                                              # decoder use of the tied weight
    g_bdec = d_z2.sum(axis=0)
    d_h = d_z2 @ w
    d_h += sparse_penalty * np.sign(h - sparse_level) / sparsity_denominator(layer, batch, patches, h.shape[1])
    hb = h.reshape(batch, patches, -1)
    diff = hb[:-1] - hb[1:]
    nrm = np.sqrt((diff ** 2).sum(axis=(1, 2)))
    gcc = diff / nrm[:, None, None] * (consecutive_penalty / (batch - 1))
    d_hb = np.zeros_like(hb)
    d_hb[:-1] += gcc
    d_hb[1:] -= gcc
    d_h += d_hb.reshape(rows, -1)

    g_ws = [None] * (layer + 1)
    g_bes = [None] * (layer + 1)
    d_hl = d_h
    for l in range(layer, -1, -1):
        d_z1 = d_hl * hs[l] * (1 - hs[l])
        g = xts[l].T @ d_z1
        g_ws[l] = g + g_w if l == layer else g
        g_bes[l] = d_z1.sum(axis=0)
        if l == 0:
            break
        d_xt = d_z1 @ ws[l].T
        if l == layer:
            d_xt = d_xt + d_labels                                # labels of layer >= 1 are the corrupted input
        # x_tilde_l = h_{l-1} * mask_l  (mask broadcast over frames)
        d_hl = (d_xt.reshape(batch, patches, -1) * masks[l][None]).reshape(rows, -1)
    return loss, (cd, cs, cc), g_ws, g_bes, g_bdec


def sgd_step(layer, x, masks, ws, b_encs, b_decs, lr=0.1, **kw):
    """One ``sess.run(train_steps[layer])`` (SDAV.py:223-226,262): returns (loss before the
    update, new ws, new b_encs, new b_decs)."""
    loss, _, g_ws, g_bes, g_bdec = loss_and_grads(layer, x, masks, ws, b_encs, b_decs[layer], **kw)
    ws = [w - lr * g_ws[l] if l <= layer else w for l, w in enumerate(ws)]
    b_encs = [b - lr * g_bes[l] if l <= layer else b for l, b in enumerate(b_encs)]
    b_decs = [b - lr * g_bdec if l == layer else b for l, b in enumerate(b_decs)]
    return loss, ws, b_encs, b_decs
