"""Randomised stress of the fp64 GEMM / convolution / similarity paths (GPU box only; not part of the test suite):
many random shapes, every result checked bit for bit against an independent route through the library --
  GEMM         LDS-DMA kernel  vs  register-staged kernel (odd row stride of A);
  convolution  implicit GEMM (+ per-frame min / max keys)  vs  im2col + GEMM (+ row min / max pass);
  similarity   whole matrix  vs  frame ranges scored alone;  arg-min filter (int8 products)  vs  fp64 Gram form.
Usage: python scripts/stress_fp64.py [seconds per family, default 60] [seed] [gemm | conv | similarity]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import deeploopcloser_amd as dlc
from deeploopcloser_amd import _lib as L

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
eng = dlc.default_engine()
rng = np.random.RandomState(seed)
g = torch.Generator(device=eng.device); g.manual_seed(seed)
rnd = lambda *shape: torch.randn(shape, generator=g, device=eng.device, dtype=torch.float64)


def gemm_case():
    m = int(rng.randint(1, 9000)); n = int(rng.randint(1, 1500)); k = int(rng.randint(1, 900))
    if rng.rand() < 0.7:
        n += n & 1; k += k & 1
    lay = L.DLC_B_KN if rng.rand() < 0.5 else L.DLC_B_NK
    act = int(rng.randint(0, 3))
    a = rnd(m, k) / max(1, k) ** 0.5
    b = rnd(k, n) if lay == L.DLC_B_KN else rnd(n, k)
    bias = rnd(n) if rng.rand() < 0.8 else None
    got = eng.gemm_bias_act(a, b, bias, act=act, blayout=lay)
    lda = k + 1 + (k & 1)                                            # an odd row stride: rows not 16-byte aligned
    wide = torch.zeros((m, lda), dtype=torch.float64, device=eng.device)
    wide[:, :k] = a
    out = torch.empty((m, n), dtype=torch.float64, device=eng.device)
    eng._check(eng.lib.dlc_gemm_bias_act(eng.ctx, L.DLC_F64, lay, act, m, n, k, wide.data_ptr(), lda, b.data_ptr(), b.stride(0),
                                          bias.data_ptr() if bias is not None else None, out.data_ptr(), n, None))
    torch.cuda.synchronize()
    assert torch.equal(out, got), ("gemm", m, n, k, lay, act)
    z = a @ (b if lay == L.DLC_B_KN else b.T) + (bias if bias is not None else 0)
    ref = torch.sigmoid(z) if act == 1 else (torch.relu(z) if act == 2 else z)
    assert float((got - ref).abs().max()) < 1e-10, ("gemm vs torch", m, n, k)
    return (m, n, k)


def conv_case():
    c = 16 * int(rng.randint(1, 9)); cout = int(rng.randint(1, 300)); kh = int(rng.choice([1, 3, 5, 7]))
    h = int(rng.randint(kh, 40)); w = int(rng.randint(kh, 40)); stride = int(rng.randint(1, 4))
    same = rng.rand() < 0.6
    if same:
        oh, ow = -(-h // stride), -(-w // stride)
        pt = max((oh - 1) * stride + kh - h, 0) // 2; pl = max((ow - 1) * stride + kh - w, 0) // 2
    else:
        oh, ow = (h - kh) // stride + 1, (w - kh) // stride + 1
        pt = pl = 0
    cap = 3000000 if rng.rand() < 0.15 else 200000                 # now and then large enough for several rounds of tiles (row-split launches)
    n = min(60000, int(rng.randint(1, max(2, cap // (oh * ow * 16)))))
    x, wk, b = rnd(n, h, w, c), rnd(kh * kh * c, cout) / (kh * c ** 0.5), rnd(cout)
    act = int(rng.choice([0, 2]))
    keys = eng.frame_minmax_keys(n)
    y = eng.conv2d(x, wk, b, kh, kh, stride, pt, pl, oh, ow, act, frame_keys=keys)
    cols = eng.im2col(x, kh, kh, stride, pt, pl, oh, ow)
    alt = eng.gemm_bias_act(cols, wk, b, act=act).reshape(y.shape)
    assert torch.equal(alt, y), ("conv", n, h, w, c, kh, cout, stride, same)
    sel = torch.arange(0, y[0].numel(), 5, device=eng.device)
    assert torch.equal(eng.quant_gather([y], sel, keys), eng.minmax_quant_gather([y], sel)), ("conv keys", n, h, w, c, kh, cout, stride, same)
    return (n, h, w, c, kh, cout, stride, same)


def sim_case():
    """whole matrix vs frame ranges scored alone, and the arg-min filter (csrc/gram_i8.hip) vs the fp64 Gram form"""
    p = int(rng.randint(1, 33)); hdim = 2 * int(rng.randint(8, 80)) if rng.rand() < 0.7 else int(rng.randint(1, 700))
    n = int(rng.randint(2, max(3, 9000 // p)))
    kind = int(rng.randint(0, 4))
    if kind == 0:
        ds = torch.rand((n, p, hdim), generator=g, device=eng.device, dtype=torch.float64)
    elif kind == 1:
        ds = torch.sigmoid(35.0 * torch.randn((n, p, hdim), generator=g, device=eng.device, dtype=torch.float64))
    elif kind == 2:
        ds = 5.0 * torch.randn((n, p, hdim), generator=g, device=eng.device, dtype=torch.float64) + 2.0
    else:                                                             # a video: every frame the one before plus a little noise
        base = torch.rand((1, p, hdim), generator=g, device=eng.device, dtype=torch.float64)
        ds = base + 1e-3 * torch.cumsum(torch.randn((n, p, hdim), generator=g, device=eng.device, dtype=torch.float64), 0)
    if n * p > 8 and rng.rand() < 0.3:                                # duplicated patches: exact ties, first index wins
        flat = ds.reshape(n * p, hdim)
        src = torch.from_numpy(rng.randint(0, n * p, size=n * p // 5 + 1)).to(eng.device)
        dst = torch.from_numpy(rng.randint(0, n * p, size=n * p // 5 + 1)).to(eng.device)
        flat[dst] = flat[src].clone()
    score = eng.distinctive_score(ds, 0.5, 0.2)
    mf, mi = eng.sdav_similarity_matrix(ds, score, 10.0, -10.0)
    # the two forms (int8 arg-min filter; fp64 Gram, DLC_SIM_FORCE_F64) give the same matrix on ANY data since round 3: both
    # hand what their products cannot decide to the same direct evaluation
    mf, mi = mf.clone(), mi.clone()
    rf, ri = eng.sdav_similarity_matrix(ds, score, 10.0, -10.0, force_f64=True)
    nan_eq = torch.equal(mf.isnan(), rf.isnan()) and torch.equal(torch.nan_to_num(mf), torch.nan_to_num(rf))
    assert nan_eq and torch.equal(mi, ri), ("filter vs fp64 Gram", n, p, hdim, kind, int((mf != rf).sum()))
    assert torch.equal(mf, mf.T) and torch.equal(mi, mi.T)
    lo = int(rng.randint(0, n - 1)); hi = int(rng.randint(lo + 2, n + 1)) if lo + 2 <= n else n
    sub, _ = eng.sdav_similarity_matrix(ds[lo:hi].contiguous(), score, 10.0, -10.0, want_int64=False)
    assert torch.equal(sub, mf[lo:hi, lo:hi]), ("similarity", n, p, hdim, lo, hi)
    return (n, p, hdim)


only = sys.argv[3] if len(sys.argv) > 3 else None
for name, fn in (("gemm", gemm_case), ("conv", conv_case), ("similarity", sim_case)):
    if only and name != only:
        continue
    t0, cnt, last = time.time(), 0, None
    while time.time() - t0 < budget:
        r = fn()
        if r != "skip":
            cnt += 1; last = r
    torch.cuda.synchronize()
    print("%s: %d random cases ok (last %s)" % (name, cnt, last), flush=True)
