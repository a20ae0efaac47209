"""Shader-clock cycles and wall time of single score-GEMM workgroups (GPU box only; experimental
builds with -DDLC_LIFE): tiles 300, 812, 1324, ... of the 1M-row launch."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
    for _ in range(5):
        eng.score_groups(q, db, k, ws)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 32)()
    raw = C.CDLL(L.LIB_PATH)
    assert raw.dlc_debug_life(buf) == 0
    print(path)
    for s in range(8):
        c0, c1, r0, r1 = buf[4 * s:4 * s + 4]
        if r1 > r0:
            cyc, us = c1 - c0, (r1 - r0) / 100.0
            print("  tile %4d: %8d cycles  %7.1f us  -> %.3f GHz   (%.0f cycles / K tile)" %
                  (300 + 512 * s, cyc, us, cyc / us / 1e3, cyc / 64))
    del db, q, ws
    eng.close()
