"""Barrier-arrival stamps of one score-GEMM workgroup (GPU box only; experimental builds with
-DDLC_STAMPS, see scripts/exp_build.sh).  Prints, for 4 consecutive K tiles, when each of the 8
waves reached each of the 4 barriers (cycles of s_memtime relative to the first arrival)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import deeploopcloser_amd as dlc
from deeploopcloser_amd import _lib as L

for path in sys.argv[1:]:
    L._lib = None
    L.LIB_PATH = os.path.abspath(path)
    dlc.engine._default.clear()
    eng = dlc.Engine(0)
    n, d, nq, k = 1_000_000, 4096, 256, 20
    db = torch.randn((n, d), device=eng.device, dtype=torch.float32).to(torch.bfloat16)
    q = torch.randn((nq, d), device=eng.device, dtype=torch.float32).to(torch.bfloat16)
    ws = torch.empty(eng.topk_workspace_bytes(nq, n, d, k), dtype=torch.uint8, device=eng.device)
    for _ in range(3):
        eng.score_groups(q, db, k, ws)          # the GEMM only: some experimental builds produce garbage scores
    torch.cuda.synchronize()
    buf = (C.c_uint * 128)()
    raw = C.CDLL(L.LIB_PATH)
    assert raw.dlc_debug_stamps(buf) == 0
    st = np.array(list(buf), dtype=np.int64).reshape(8, 16)
    st = (st - st.min()) & 0xffffffff
    print(path)
    print("stamp   " + " ".join("w%d    " % w for w in range(8)) + "  | g0 mean  g1 mean   g1-g0   g0 next-g1")
    for b in range(16):
        col = st[:, b]
        g0, g1 = col[:4].mean(), col[4:].mean()
        line = "%2d     " % b + " ".join("%6d" % v for v in col) + "  | %7.0f %7.0f %7.0f" % (g0, g1, g1 - g0)
        if b + 1 < 16:
            line += " %7.0f" % (st[:4, b + 1].mean() - g1)
        print(line)
    del db, q
    eng.close()
