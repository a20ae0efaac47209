"""Board power and clocks while the score GEMM of each given library loops (GPU box only).
Samples `rocm-smi --showpower --showclocks` from a child process once a second."""
import os, re, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deeploopcloser_amd as dlc
from deeploopcloser_amd import _lib as L

def sample(out, stop):
    while not stop.is_set():
        try:
            txt = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
            pw = re.search(r"Power \(W\):\s*([0-9.]+)", txt)
            sc = re.search(r"sclk clock level:.*?\((\d+)Mhz\)", txt)
            mc = re.search(r"mclk clock level:.*?\((\d+)Mhz\)", txt)
            out.append((float(pw.group(1)) if pw else -1, int(sc.group(1)) if sc else -1, int(mc.group(1)) if mc else -1))
        except Exception as e:                       # noqa
            out.append((-1, -1, -1))
        stop.wait(1.0)

for path in sys.argv[1:]:
    L._lib = None
    L.LIB_PATH = os.path.abspath(path)
    dlc.engine._default.clear()
    eng = dlc.Engine(0)
    n, d, nq, k = 1_000_000, 4096, int(os.environ.get("DLC_EXP_Q", "256")), 20
    db = torch.randn((n, d), device=eng.device, dtype=torch.float32).to(torch.bfloat16)
    q = torch.randn((nq, d), device=eng.device, dtype=torch.float32).to(torch.bfloat16)
    ws = torch.empty(eng.topk_workspace_bytes(nq, n, d, k), dtype=torch.uint8, device=eng.device)
    for _ in range(20):
        eng.score_groups(q, db, k, ws)
    torch.cuda.synchronize()
    out, stop = [], threading.Event()
    th = threading.Thread(target=sample, args=(out, stop))
    th.start()
    t0 = time.perf_counter()
    iters = 0
    while time.perf_counter() - t0 < 7.0:
        for _ in range(100):
            eng.score_groups(q, db, k, ws)
        torch.cuda.synchronize()
        iters += 100
    dt = (time.perf_counter() - t0) / iters * 1e3
    stop.set()
    th.join()
    good = [o for o in out[1:] if o[0] > 0]
    print("Q=%d %-32s %.3f ms/launch back-to-back; samples (W, sclk MHz, mclk MHz): %s" % (nq, os.path.basename(path), dt, good), flush=True)
    del db, q, ws
    eng.close()
    time.sleep(3)
