"""The training-step oracle (oracle/sdav_train.py): analytic gradients against central finite
differences of the restated loss (SDAV.py:171-186), for the trained layer AND the earlier
layers the loss reaches through."""
import numpy as np
import pytest

from oracle import sdav_train as ot
from oracle.tensor_ops import corruption_mask


def setup(layer, seed=0, batch=3, patches=4, dims=(6, 5, 7, 4)):
    rng = np.random.RandomState(seed)
    x = rng.uniform(0, 1, size=(batch, patches, dims[0]))
    ws = [rng.standard_normal((a, b)) * 0.5 for a, b in zip(dims[:-1], dims[1:])]
    b_encs = [rng.standard_normal(b) * 0.1 for b in dims[1:]]
    b_dec = rng.standard_normal(dims[layer]) * 0.1
    masks = [corruption_mask((patches, d), 0.3, rng) for d in dims[:-1]]
    return x, masks, ws, b_encs, b_dec


def num_grad(f, arr, eps=1e-6):
    g = np.zeros_like(arr)
    it = np.nditer(arr, flags=["multi_index"])
    for _ in it:
        i = it.multi_index
        old = arr[i]
        arr[i] = old + eps
        fp = f()
        arr[i] = old - eps
        fm = f()
        arr[i] = old
        g[i] = (fp - fm) / (2 * eps)
    return g


@pytest.mark.parametrize("layer", [0, 1, 2])
def test_analytic_gradients_match_finite_differences(layer):
    x, masks, ws, b_encs, b_dec = setup(layer)
    loss, parts, g_ws, g_bes, g_bd = ot.loss_and_grads(layer, x, masks, ws, b_encs, b_dec)
    f = lambda: ot.loss_and_grads(layer, x, masks, ws, b_encs, b_dec)[0]
    assert np.isfinite(loss) and abs(loss - (parts[0] + 1.0 * parts[1] + 0.2 * parts[2])) < 1e-12
    for l in range(layer + 1):
        np.testing.assert_allclose(g_ws[l], num_grad(f, ws[l]), rtol=2e-5, atol=2e-8)
        np.testing.assert_allclose(g_bes[l], num_grad(f, b_encs[l]), rtol=2e-5, atol=2e-8)
    np.testing.assert_allclose(g_bd, num_grad(f, b_dec), rtol=2e-5, atol=2e-8)


def test_loss_pieces_and_sgd_step():
    x, masks, ws, b_encs, b_dec = setup(0)
    xt, h, y = ot.forward_layer(x, masks[0], ws[0], b_encs[0], b_dec)
    loss, cd, cs, cc = ot.layer_loss(x.reshape(12, 6), h, y, 3, 4, layer=0)
    # softmax cross entropy with sigmoid outputs as logits and x as (unnormalised) labels (:172)
    lsm = y - np.log(np.exp(y).sum(1, keepdims=True))
    assert abs(cd - np.mean(-(x.reshape(12, 6) * lsm).sum(1))) < 1e-12
    # layer 0: h is [B,P,N] in the reference's graph, tf.norm(axis=1) sums over the P patches (:174)
    hb = h.reshape(3, 4, -1)
    assert abs(cs - np.mean(np.abs(hb - 0.05).sum(1))) < 1e-12
    assert abs(cs - np.abs(h - 0.05).sum() / (3 * h.shape[1])) < 1e-12
    # layers >= 1: h is [B*P,N], axis 1 is the unit axis
    cs1 = ot.layer_loss(x.reshape(12, 6), h, y, 3, 4, layer=1)[2]
    assert abs(cs1 - np.mean(np.abs(h - 0.05).sum(1))) < 1e-12 and abs(cs1 / cs - h.shape[1] / 4) < 1e-9
    assert abs(cc - np.mean([np.linalg.norm(hb[0] - hb[1]), np.linalg.norm(hb[1] - hb[2])])) < 1e-12
    b_decs = [b_dec] + [np.zeros(5), np.zeros(7)]
    l0, w1, be1, bd1 = ot.sgd_step(0, x, masks, ws, b_encs, b_decs, lr=0.1)
    l1 = ot.loss_and_grads(0, x, masks, w1, be1, bd1[0])[0]
    assert l1 < l0 and np.array_equal(w1[1], ws[1]) and not np.array_equal(w1[0], ws[0])
    with pytest.raises(ValueError):
        ot.loss_and_grads(0, x[:1], masks, ws, b_encs, b_dec)


def test_corruption_mask_counts():
    rng = np.random.RandomState(0)
    m = corruption_mask((30, 1681), 0.3, rng)
    assert m.shape == (30, 1681) and int((m == 0).sum()) == int(np.round(30 * 1681 * 0.3))
    assert corruption_mask((30, 1681), 0, rng).all()


@pytest.mark.parametrize("layer", [0, 1, 2])
def test_analytic_gradients_match_torch_autograd(layer):
    """Independent check: the same graph written with torch ops (SDAV.py:126-159 forward, :171-186
    loss, tied decoder :192-216) and differentiated by autograd -- including the gradient that
    softmax_cross_entropy_with_logits_v2 sends into its labels for layers >= 1."""
    import torch
    x, masks, ws, b_encs, b_dec = setup(layer, seed=3)
    loss, _, g_ws, g_bes, g_bd = ot.loss_and_grads(layer, x, masks, ws, b_encs, b_dec)
    tw = [torch.tensor(w, requires_grad=True) for w in ws]
    tb = [torch.tensor(b, requires_grad=True) for b in b_encs]
    tbd = torch.tensor(b_dec, requires_grad=True)
    batch, patches, _ = x.shape
    cur = torch.tensor(x)
    for l in range(layer + 1):
        xt = (cur * torch.tensor(masks[l])[None]).reshape(batch * patches, -1)
        h = torch.sigmoid(xt @ tw[l] + tb[l])
        cur = h.reshape(batch, patches, -1)
    y = torch.sigmoid(h @ tw[layer].T + tbd)                                 # tied decoder weights
    labels = torch.tensor(x).reshape(batch * patches, -1) if layer == 0 else xt
    cd = (-(labels * torch.log_softmax(y, dim=1)).sum(1)).mean()
    hb = h.reshape(batch, patches, -1)
    # tf.norm(h - s, axis=1, ord=1): h is 3-D [B,P,N] at layer 0 (axis 1 = patches), 2-D afterwards
    cs = (hb - 0.05).abs().sum(1).mean() if layer == 0 else (h - 0.05).abs().sum(1).mean()
    cc = ((hb[:-1] - hb[1:]) ** 2).sum((1, 2)).sqrt().mean()
    tl = cd + 1.0 * cs + 0.2 * cc
    tl.backward()
    assert abs(float(tl.detach()) - loss) < 1e-12
    for l in range(layer + 1):
        np.testing.assert_allclose(g_ws[l], tw[l].grad.numpy(), rtol=1e-10, atol=1e-13)
        np.testing.assert_allclose(g_bes[l], tb[l].grad.numpy(), rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(g_bd, tbd.grad.numpy(), rtol=1e-10, atol=1e-13)
