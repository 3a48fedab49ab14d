#!/usr/bin/env python3
"""Order of global loads (L), stores (S), LDS ops (d), s_waitcnt vmcnt(n) (Wn), MFMAs (M) and branches (b) in a kernel's
device assembly -- the view that shows a loop whose loads are closed one by one with s_waitcnt vmcnt(0) (several memory
round trips per step where one would do).  Developer tool.
usage: isa_waits.py FILE.hip KERNEL_SUBSTRING [KERNEL_SUBSTRING ...]"""
import re, subprocess, sys
src, names = sys.argv[1], sys.argv[2:]
asm = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize" if "blend" in src else "-O3",
                      "--cuda-device-only", "-S", src, "-o", "-"], capture_output=True, text=True).stdout
lines = asm.splitlines()
for i, l in enumerate(lines):
    m = re.match(r"^(_Z\w+):", l)
    if not m or not any(n in m.group(1) for n in names):
        continue
    end = next(j for j in range(i, len(lines)) if "s_endpgm" in lines[j])
    seq = []
    for t in (x.strip() for x in lines[i:end]):
        if not t or t.startswith((";", ".")) and not t.startswith(".LBB"):
            continue
        op = t.split()[0]
        if t.startswith(".LBB"): seq.append("|")
        elif op.startswith(("global_load", "buffer_load", "flat_load")): seq.append("L")
        elif op.startswith(("global_store", "buffer_store", "flat_store")): seq.append("S")
        elif op.startswith("global_atomic"): seq.append("A")
        elif op.startswith("s_waitcnt") and "vmcnt" in t: seq.append("W" + re.search(r"vmcnt\((\d+)\)", t).group(1))
        elif op.startswith("v_mfma"): seq.append("M")
        elif op.startswith("s_barrier"): seq.append("B")
        elif op.startswith("s_cbranch") or op == "s_branch": seq.append("b")
    out, prev, n = [], None, 0
    for x in seq + [None]:
        if x == prev:
            n += 1
            continue
        if prev:
            out.append(f"{prev}x{n}" if n > 1 else prev)
        prev, n = x, 1
    print(m.group(1)[:70], "\n   ", " ".join(out), "\n")
