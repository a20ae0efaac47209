"""Latency of the reference-semantics streaming query (SimilarityStream.query) at the reference's size: one new frame
against 1062 resident frames of 30 x 2500 descriptors.  GPU box only."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 2:
    import deeploopcloser_amd._lib as L
    L.LIB_PATH = os.path.abspath(sys.argv[2])
import deeploopcloser_amd as dlc
eng = dlc.default_engine()
g = torch.Generator(device=eng.device); g.manual_seed(3)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1063
ds = torch.sigmoid(35.0 * torch.randn((n, 30, 2500), generator=g, device=eng.device, dtype=torch.float64)).clamp_(0, 1)
score = eng.distinctive_score(ds, 0.5, 0.2)
st = dlc.SimilarityStream(score, capacity=n)
st.append(ds)
torch.cuda.synchronize()
for f in (n - 1, n // 2):
    st.query(f); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        st.query(f)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    mb = f * 30 * 7680 / 1e6
    print("query of frame %d against %d older frames: %.1f us per query (%.0f MB of panel: %.2f TB/s), direct evaluations %d"
          % (f, f, us, mb, mb / us, int(st.stats[0])))
for cnt in (8, 32):                        # a batch of frames that arrived together: one pair of launches
    st.query_batch(n - cnt, cnt); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        st.query_batch(n - cnt, cnt)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print("batch of %d queries (frames %d .. %d): %.1f us per batch = %.1f us per query, direct evaluations %d"
          % (cnt, n - cnt, n - 1, us, us / cnt, int(st.stats[0])))
t0 = time.perf_counter()
st2 = dlc.SimilarityStream(score, capacity=n)
for f in range(64):
    st2.query_and_insert(ds[f])
torch.cuda.synchronize()
print("64 x query_and_insert from empty: %.2f ms per frame (host-paced)" % ((time.perf_counter() - t0) / 64 * 1e3))
mf, _ = eng.sdav_similarity_matrix(ds, score, 10.0, -10.0, want_int64=False)
row = st.query(n - 1)
print("row == matrix column:", bool(torch.equal(torch.nan_to_num(row, posinf=1e300), torch.nan_to_num(mf[:n - 1, n - 1], posinf=1e300))))
