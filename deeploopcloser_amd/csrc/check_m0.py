#!/usr/bin/env python3
"""Build check: gram_i8_kernel's LDS-DMA pieces overwrite M0 without saving it, which is safe only while NOTHING
else in that kernel touches M0.  Disassemble the object and fail if an instruction other than the pieces' own
`s_mov_b32 m0, ...` (always followed by s_nop + global_load_lds) names m0.
Second check (r03): the kernel's MFMAs are inline asm on FIXED accumulation registers a0 .. a191, opaque to hipcc's hazard
recognizer -- an accumulator that hipcc read or moved on its own would carry no wait states behind the MFMA that wrote it.
Fail unless the only v_accvgpr_* instructions are the 192 zero writes in front of the k loop and the 192 reads of the
epilogue, all BEHIND the last MFMA's two `s_nop 15`.   usage: check_m0.py build/gram_i8.o"""
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
acc = [(i, l) for i, l in enumerate(lines) if l.startswith("v_accvgpr")]
mfma = [i for i, l in enumerate(lines) if l.startswith("v_mfma")]
assert mfma, "no MFMA in gram_i8_kernel"
writes = [(i, l) for i, l in acc if l.startswith("v_accvgpr_write_b32")]
reads = [(i, l) for i, l in acc if l.startswith("v_accvgpr_read_b32")]
other = [l for i, l in acc if not l.startswith(("v_accvgpr_write_b32", "v_accvgpr_read_b32"))]
problems = []
if other:
    problems.append("accumulator moves: " + "; ".join(other[:4]))
if len(writes) != 192 or any(not re.match(r"v_accvgpr_write_b32 a\d+, 0$", l) for _, l in writes) or any(i > mfma[0] for i, _ in writes):
    problems.append("%d v_accvgpr_write (want 192 zero writes in front of the first MFMA)" % len(writes))
if len(reads) != 192 or any(i < mfma[-1] for i, _ in reads):
    problems.append("%d v_accvgpr_read (want 192, all behind the last MFMA)" % len(reads))
else:
    between = lines[mfma[-1] + 1:reads[0][0]]
    if sum(l == "s_nop 15" for l in between) < 2:
        problems.append("no two `s_nop 15` between the last MFMA and the first accumulator read")
if problems:
    sys.exit("gram_i8_kernel: hipcc touched the fixed accumulators:\n  " + "\n  ".join(problems))
shutil.rmtree(tmp, ignore_errors=True)
print("check_m0: %d MFMAs on fixed accumulators, 192 zero writes in front of them and 192 reads behind them" % len(mfma))
print("check_m0: %d LDS-DMA pieces in gram_i8_kernel, no other M0 access" % sum("global_load_lds_dwordx4" in l for l in lines))
