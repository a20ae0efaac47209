import os, sys, time, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import deeploopcloser_amd as dlc
eng = dlc.default_engine()
print("THP:", open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip(), "| defrag:", open("/sys/kernel/mm/transparent_hugepage/defrag").read().strip())
print("cpus:", os.cpu_count(), "affinity:", len(os.sched_getaffinity(0)))
h = torch.rand((1063 * 30, 2500), dtype=torch.float64, device=eng.device)
libc = ctypes.CDLL("libc.so.6", use_errno=True)
MADV_HUGEPAGE = 14
def t(fn, reps=4):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e3
out = np.empty((1063 * 30, 2500)); out[:] = 0
print("download into a pre-faulted array: %.1f ms" % t(lambda: eng.download(h, out=out)))
print("download into a fresh np.empty:    %.1f ms" % t(lambda: eng.download(h)))
def huge():
    o = np.empty((1063 * 30, 2500))
    a = o.ctypes.data
    lo = (a + (2 << 20) - 1) & ~((2 << 20) - 1)
    rc = libc.madvise(ctypes.c_void_p(lo), ctypes.c_size_t((a + o.nbytes - lo) & ~((2 << 20) - 1)), MADV_HUGEPAGE)
    eng.download(h, out=o)
    return rc
print("fresh + MADV_HUGEPAGE:             %.1f ms (rc %s)" % (t(huge), huge()))
def touch_only():
    o = np.empty((1063 * 30, 2500)); o[::512] = 0     # one write per page
print("first touch alone (single thread, one write per 4 KiB): %.1f ms" % t(touch_only))
for th in (4, 8, 16):
    eng._check(eng.lib.dlc_set_host_threads(eng.ctx, th))
    print("threads %d: pre-faulted %.1f ms, fresh %.1f ms" % (th, t(lambda: eng.download(h, out=out)), t(lambda: eng.download(h))))
