"""SDAV training step on the GPU (dlc_sdav_train_step) against the finite-difference-pinned
oracle (oracle/sdav_train.py): same batch, same masks, same parameters -> same loss and same
updated parameters.  fp64 on both sides: only the summation order differs (tolerance 1e-9 rel)."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def close(a, b, rel=1e-9):
    a, b = np.asarray(a), np.asarray(b)
    return np.abs(a - b).max() <= rel * max(1.0, np.abs(b).max())


@pytest.mark.parametrize("layer", [0, 1, 2, 3, 4])
def test_train_step_small_vs_oracle(layer):
    """One SGD step of layer `layer` of a small five-layer network against oracle/sdav_train.py: loss parts and every
    parameter the loss reaches.  Layers 3 and 4 take the step's other route for the encoder biases of layers 2 .. layer - 1
    (their dz1 buffer is reused before the update kernel runs: a column-sum launch of their own)."""
    import deeploopcloser_amd as dlc
    from oracle import sdav_train as ot
    from oracle.tensor_ops import corruption_mask
    eng = dlc.default_engine()
    rng = np.random.RandomState(layer)
    batch, patches, dims = 4, 5, (37, 21, 21, 21, 21, 21)
    x = rng.uniform(0, 1, size=(batch, patches, dims[0]))
    ws = [rng.standard_normal((a, b)) * 0.4 for a, b in zip(dims[:-1], dims[1:])]
    bes = [rng.standard_normal(b) * 0.1 for b in dims[1:]]
    bds = [rng.standard_normal(a) * 0.1 for a in dims[:-1]]
    masks = [corruption_mask((patches, d), 0.3, rng) for d in dims[:-1]]
    want_loss, (cd, cs, cc), _, _, _ = ot.loss_and_grads(layer, x, masks, ws, bes, bds[layer])
    _, w1, be1, bd1 = ot.sgd_step(layer, x, masks, ws, bes, bds, lr=0.1)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(eng.device)
    W, BE, BD, M = [dev(w) for w in ws], [dev(b) for b in bes], [dev(b) for b in bds], [dev(m) for m in masks]
    loss = torch.empty(4, dtype=torch.float64, device=eng.device)
    eng.sdav_train_step(layer, dev(x.reshape(-1, dims[0])), batch, patches, M, W, BE, BD[layer], 0.05, 1.0, 0.2, 0.1,
                        loss_out=loss)
    got = loss.cpu().numpy()
    assert close(got, [want_loss, cd, cs, cc])
    for l in range(len(ws)):
        assert close(W[l].cpu().numpy(), w1[l]) and close(BE[l].cpu().numpy(), be1[l])
        if l > layer:
            assert np.array_equal(W[l].cpu().numpy(), ws[l])                   # untouched
    assert close(BD[layer].cpu().numpy(), bd1[layer])


def test_sdav_train_step_real_shape_and_fit_surface(tmp_path):
    import deeploopcloser_amd as dlc
    from oracle import sdav_train as ot
    rng = np.random.RandomState(3)
    net = dlc.SDAV(seed=6, weight_scale="fan_in")
    x = rng.uniform(0, 1, size=(3, 30, 1681))
    ws, bs = net.get_weights()
    bds = [b.cpu().numpy() for b in net._biases_dec]
    masks = [net._mask(l).cpu().numpy() for l in range(2)]
    assert masks[0].shape == (30, 1681) and int((masks[0] == 0).sum()) == int(np.round(30 * 1681 * 0.3))
    want_loss, w1, be1, bd1 = ot.sgd_step(1, x, masks, ws, bs, bds, lr=net.learning_rate)
    loss = net.train_step(1, x, masks=masks)
    assert close(loss[0].item(), want_loss)
    w_new, b_new = net.get_weights()
    for l in range(5):
        assert close(w_new[l], w1[l]) and close(b_new[l], be1[l])
    assert close(net._biases_dec[1].cpu().numpy(), bd1[1]) and net.global_step == 1
    # the loss goes down on repeated steps of one layer (same batch), as SDAV.fit drives it
    net.epochs = 3
    l0 = float(net.train_step(0, x)[0])
    for _ in range(5):
        l1 = float(net.train_step(0, x)[0])
    assert l1 < l0
    # train.py surface: get_dataset / fit_dataset on the reference's frames, checkpoint + reload
    net2 = dlc.SDAV(seed=6, weight_scale="fan_in")
    net2.epochs, net2.default_batch_size = 1, 3
    net2.checkpoint_file = str(tmp_path / "checkpoint_file")
    ds = net2.get_dataset(os.path.join(GOLDEN, "frames", "*.ppm"))
    net2.fit_dataset(ds)
    assert net2.global_step == 5
    ck = sorted(glob.glob(str(tmp_path / "checkpoint_file-*.npz")))
    assert len(ck) == 5
    net3 = dlc.SDAV(seed=0)
    net3.load_weights(ck[-1])
    assert net3.global_step == 5 and torch.equal(net3._weights[4], net2._weights[4])
    assert np.array_equal(net3.transform(x), net2.transform(x))
    with pytest.raises(ValueError):
        net2.train_step(0, x[:1])
    net2.fit(x)                                    # 5 layers x epochs steps on one batch
    assert net2.global_step == 10


@pytest.mark.parametrize("layer", [0, 2])
def test_train_steps_graph_replay_equals_eager_steps(layer):
    """SDAV.train_steps (the inner loop of fit / fit_dataset as ONE captured HIP graph replayed per step, masks redrawn in
    place) == the same number of train_step calls with the same mask draws: same parameters, same losses, bit for bit --
    also run after run (the frame norms are summed in a fixed order), inside latency mode (split-K scratch captured) and
    after the parameters are replaced (the capture is redone)."""
    import deeploopcloser_amd as dlc
    rng = np.random.RandomState(11)
    x = rng.uniform(0, 1, size=(10, 30, 1681))
    eng = dlc.default_engine()

    def run(graph, latency, n=6):
        net = dlc.SDAV(seed=4, weight_scale="fan_in")
        ctx = eng.latency_mode() if latency else __import__("contextlib").nullcontext()
        with ctx:
            if graph:
                loss = net.train_steps(layer, x, n).clone()
            else:
                xd = eng.to_device(x, torch.float64)
                for _ in range(n):
                    loss = net.train_step(layer, xd)            # draws its masks from the same generator, in the same order
        torch.cuda.synchronize()
        assert net.global_step == n
        return [w.clone() for w in net._weights], [b.clone() for b in net._biases], net._biases_dec[layer].clone(), loss.clone()

    for latency in (False, True):
        a, b, c = run(True, latency), run(False, latency), run(True, latency)
        for got in (a, c):
            for l in range(5):
                assert torch.equal(got[0][l], b[0][l]) and torch.equal(got[1][l], b[1][l]), (latency, l)
            assert torch.equal(got[2], b[2]) and torch.equal(got[3], b[3])
    # parameters replaced between calls: the graph must not keep writing the old tensors
    net = dlc.SDAV(seed=4, weight_scale="fan_in")
    net.train_steps(layer, x, 4)
    ws, bs = net.get_weights()
    net.set_weights([w * 0.5 for w in ws], bs)
    before = [w.clone() for w in net._weights]
    net.train_steps(layer, x, 4)
    ref = dlc.SDAV(seed=4, weight_scale="fan_in")
    ref.set_weights([w * 0.5 for w in ws], bs)
    ref._biases_dec = [b.clone() for b in net._biases_dec]
    assert not torch.equal(net._weights[layer], before[layer])
    assert net.global_step == 8


def test_random_mask_kernel_counts_and_uniformity():
    """dlc_random_mask_f64 (TensorflowWrapper.py:148-156): the zero count is EXACT for every size and level, a draw is a
    function of (seed, counter), different counters give different masks, and every position is zero with the same
    frequency (400 draws of 3000 elements at level 0.3: per-position frequency within 5 sigma of 0.3, mean exact)."""
    import deeploopcloser_amd as dlc
    eng = dlc.default_engine()
    for n, nz in ((1, 0), (1, 1), (7, 3), (1024, 1024), (1025, 0), (50430, 15129), (75000, 22500), (200001, 1), (3000, 2999),
                  (300000, 90000), (1000000, 300000)):       # (the window's list holds the first, overflows on the second)
        m = torch.empty(n, dtype=torch.float64, device=eng.device)
        eng.random_mask(m, nz, seed=5, counter=n)
        assert int((m == 0).sum()) == nz and int((m == 1).sum()) == n - nz, (n, nz)
        m2 = torch.empty_like(m)
        eng.random_mask(m2, nz, seed=5, counter=n)
        assert torch.equal(m, m2)
        if 0 < nz < n and n > 100:
            eng.random_mask(m2, nz, seed=5, counter=n + 1)
            assert not torch.equal(m, m2)
            eng.random_mask(m2, nz, seed=6, counter=n)
            assert not torch.equal(m, m2)
    n, nz, draws = 3000, 900, 400
    acc = torch.zeros(n, dtype=torch.float64, device=eng.device)
    m = torch.empty(n, dtype=torch.float64, device=eng.device)
    for c in range(draws):
        eng.random_mask(m, nz, seed=77, counter=c)
        acc += 1.0 - m
    freq = (acc / draws).cpu().numpy()
    assert abs(freq.mean() - 0.3) < 1e-12
    sigma = np.sqrt(0.3 * 0.7 / draws)
    assert np.abs(freq - 0.3).max() < 5 * sigma, np.abs(freq - 0.3).max() / sigma
    # neighbouring positions are not correlated (a hash of the index, not a stride pattern)
    net = dlc.SDAV(seed=1)
    a, b = net._mask(0), net._mask(0)
    assert a.shape == (30, 1681) and int((a == 0).sum()) == int(np.round(30 * 1681 * 0.3)) and not torch.equal(a, b)
    z = (a.view(-1) == 0).double()
    corr = float(((z[:-1] - 0.3) * (z[1:] - 0.3)).mean() / (0.3 * 0.7))
    assert abs(corr) < 0.02, corr
