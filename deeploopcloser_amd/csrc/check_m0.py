#!/usr/bin/env python3
"""Build check: gram_i8_kernel's LDS-DMA pieces overwrite M0 without saving it, which is safe only while NOTHING
else in that kernel touches M0.  Disassemble the object and fail if an instruction other than the pieces' own
`s_mov_b32 m0, ...` (always followed by s_nop + global_load_lds) names m0.   usage: check_m0.py build/gram_i8.o"""
import re
import subprocess
import sys

import glob
import os
import shutil
import tempfile

obj = os.path.abspath(sys.argv[1])
objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
# the device code object is bundled in the host object: `llvm-objdump --offloading` writes the bundles next to its input
tmp = tempfile.mkdtemp()
shutil.copy(obj, os.path.join(tmp, "k.o"))
subprocess.run([objdump, "--offloading", os.path.join(tmp, "k.o")], capture_output=True, text=True, check=True)
dev = next(iter(glob.glob(os.path.join(tmp, "k.o.*gfx950*"))), None)
assert dev, "no gfx950 code object in " + obj
asm = subprocess.run([objdump, "-d", dev], capture_output=True, text=True, check=True).stdout
in_kernel, lines = False, []
for ln in asm.splitlines():
    if re.match(r"^[0-9a-f]+ <.*>:", ln):
        in_kernel = "gram_i8_kernel" in ln
        continue
    if in_kernel:
        lines.append(ln.split("//")[0].strip())
assert any("global_load_lds_dwordx4" in l for l in lines), "gram_i8_kernel not found in " + obj
bad = []
for i, l in enumerate(lines):
    if not re.search(r"\bm0\b", l):
        continue
    ok = re.match(r"s_mov_b32 m0, ", l) and "s_nop" in lines[i + 1] and "global_load_lds_dwordx4" in lines[i + 2]
    if not ok:
        bad.append(l)
if bad:
    sys.exit("gram_i8_kernel touches M0 outside its LDS-DMA pieces:\n  " + "\n  ".join(bad[:10]))
shutil.rmtree(tmp, ignore_errors=True)
print("check_m0: %d LDS-DMA pieces in gram_i8_kernel, no other M0 access" % sum("global_load_lds_dwordx4" in l for l in lines))
