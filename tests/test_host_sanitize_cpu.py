"""The library's host-side threads under ThreadSanitizer and AddressSanitizer + UBSan (sanitizers belong on the
CPU build: the GPU pool refuses them).  csrc/host_staging_impl.h -- the copy pool (worker threads, a generation counter, two
condition variables), the pinned staging ring's bookkeeping and the per-context host lock -- is compiled with plain g++
against tests/host_sanitize/hip_stub.h, whose "DMA engine" is a thread per stream, and driven by tests/host_sanitize/driver.cpp:
10 000 round trips of random sizes from two caller threads through ONE context while a third keeps changing
dlc_set_host_threads.  Clean = exit code 0, every byte compared equal, no sanitizer report."""
import os
import subprocess

from conftest import ROOT

CSRC = os.path.join(ROOT, "deeploopcloser_amd", "csrc")


def test_host_threads_are_clean_under_tsan_and_asan():
    out = subprocess.run(["make", "-C", CSRC, "host-sanitize"], capture_output=True, text=True, timeout=900,
                         env=dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=66", ASAN_OPTIONS="detect_leaks=1"))
    text = out.stdout + out.stderr
    assert out.returncode == 0, text[-4000:]
    assert text.count("host_sanitize: 10000 round trips, 0 bad") == 2, text[-2000:]
    for needle in ("WARNING: ThreadSanitizer", "ERROR: AddressSanitizer", "runtime error:", "LeakSanitizer"):
        assert needle not in text, text[-4000:]


def test_the_product_and_the_sanitizer_build_share_one_source():
    """host_staging.hip is nothing but dlc_internal.h + the header the sanitizer build compiles."""
    src = open(os.path.join(CSRC, "host_staging.hip")).read()
    code = [l for l in src.splitlines() if l.strip() and not l.strip().startswith("//")]
    assert code == ['#include "dlc_internal.h"', '#include "host_staging_impl.h"']
