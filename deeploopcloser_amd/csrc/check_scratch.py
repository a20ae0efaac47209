#!/usr/bin/env python3
"""Build check: no kernel of the library may use scratch memory (a private segment) -- the 256-thread
selection kernels are register-capped to share a CU with the score GEMM, the MFMA kernels pin their accumulators, and a
spill in any of them is a silent slowdown.  Reads .private_segment_fixed_size of every kernel from the gfx950 code
object's metadata notes.   usage: check_scratch.py build/*.o"""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin/"
bad, total = [], 0
for obj in sys.argv[1:]:
    tmp = tempfile.mkdtemp()
    try:
        shutil.copy(obj, os.path.join(tmp, "k.o"))
        subprocess.run([LLVM + "llvm-objdump", "--offloading", os.path.join(tmp, "k.o")], capture_output=True, text=True, check=True)
        dev = next(iter(glob.glob(os.path.join(tmp, "k.o.*gfx950*"))), None)
        if dev is None:
            continue                                            # a host-only translation unit
        notes = subprocess.run([LLVM + "llvm-readelf", "--notes", dev], capture_output=True, text=True, check=True).stdout
        name = None
        for ln in notes.splitlines():
            m = re.match(r"\s*\.name:\s+(\S+)", ln)
            if m:
                name = m.group(1)
            m = re.match(r"\s*\.private_segment_fixed_size:\s+(\d+)", ln)
            if m:
                total += 1
                if int(m.group(1)) != 0:
                    bad.append((os.path.basename(obj), name, int(m.group(1))))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
if bad:
    for o, n, b in bad:
        print("check_scratch: %s: kernel with %d bytes of scratch (name follows the size in the notes: %s)" % (o, b, n))
    sys.exit(1)
print("check_scratch: %d kernels, none with a private segment" % total)
