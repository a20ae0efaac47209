# single-frame / few-frame encode latency in latency mode: the library as built against csrc/exp_build/libdlc_base.so (r05's GEMM plans)
R=$GRAFT_REPO_ROOT
python3 - <<'PY'
import os, sys, time
R = os.environ["GRAFT_REPO_ROOT"]; sys.path.insert(0, R)
import torch
def run(lib):
    import importlib
    for m in [k for k in sys.modules if k.startswith("deeploopcloser_amd")]:
        del sys.modules[m]
    import deeploopcloser_amd._lib as L
    if lib: L.LIB_PATH = lib
    import deeploopcloser_amd as dlc
    eng = dlc.default_engine()
    net = dlc.SDAV(seed=1)
    out = []
    with eng.latency_mode():
        for b in (1, 2, 4, 8, 16, 20, 32, 64):
            x = torch.rand((b, 30, 1681), dtype=torch.float64, device=eng.device)
            for _ in range(5): net.transform_tensor(x)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30): h = net.transform_tensor(x)
            e1.record(); torch.cuda.synchronize()
            out.append((b, e0.elapsed_time(e1) / 30 * 1e3, float(h.sum())))
    return out
new = run(None)
base = run(os.path.join(R, "deeploopcloser_amd/csrc/exp_build/libdlc_base.so"))
for (b, tn, sn), (_, tb, sb) in zip(new, base):
    print("SDAV.transform_tensor, %3d frames, latency mode: %.0f us (r05 plans: %.0f us); sum rel diff %.2g" % (b, tn, tb, abs(sn - sb) / abs(sb)))
PY
