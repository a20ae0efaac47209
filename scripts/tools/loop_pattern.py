#!/usr/bin/env python3
"""Instruction-class pattern of one kernel's body (run-length encoded): what the compiler made of a hand-ordered main loop.
usage: loop_pattern.py build/x.o kernel_substring [max chars]"""
import glob, os, re, shutil, subprocess, sys, tempfile
obj, pat = sys.argv[1], sys.argv[2]
mx = int(sys.argv[3]) if len(sys.argv) > 3 else 4000
L = "/opt/rocm/lib/llvm/bin/"
tmp = tempfile.mkdtemp()
shutil.copy(obj, tmp + "/k.o")
subprocess.run([L + "llvm-objdump", "--offloading", tmp + "/k.o"], capture_output=True, check=True)
dev = glob.glob(tmp + "/k.o.*gfx950*")[0]
txt = subprocess.run([L + "llvm-objdump", "-d", dev], capture_output=True, text=True, check=True).stdout
for m in re.finditer(r"^[0-9a-f]+ <(.*)>:\n((?:.*\n)*?)(?=^[0-9a-f]+ <|\Z)", txt, re.M):
    if pat not in m.group(1):
        continue
    lines = [l.split("//")[0].strip() for l in m.group(2).splitlines()]
    def cls(l):
        op = l.split()[0] if l else ""
        for k, v in (("v_mfma", "MFMA"), ("ds_read", "DSR"), ("ds_write", "DSW"), ("global_load_lds", "DMA"), ("s_barrier", "BARRIER")):
            if op.startswith(k):
                return v
        if op.startswith(("s_waitcnt", "s_cbranch", "s_branch", "s_setprio", "global_", "scratch_", "buffer_")):
            return l
        return "VALU" if op.startswith("v_") else ("SALU" if op.startswith("s_") else op)
    out, prev, cnt = [], None, 0
    for l in lines:
        c = cls(l)
        if c == prev:
            cnt += 1
        else:
            if prev is not None:
                out.append("%s%s" % (prev, "x%d" % cnt if cnt > 1 else ""))
            prev, cnt = c, 1
    out.append("%s%s" % (prev, "x%d" % cnt if cnt > 1 else ""))
    s = " ".join(out)
    i0 = s.find("BARRIER")
    print(m.group(1)[:70]); print(s[i0:i0 + mx]); print()
    break
shutil.rmtree(tmp, ignore_errors=True)
