#!/usr/bin/env python3
"""Instruction mix of a kernel's loops from the device assembly (hipcc --cuda-device-only -S): counts per class for
every basic block between a label and the backward branch that closes a loop, so that the PMC instruction counters
(SQ_INSTS_VALU & co) can be set against the static inner-loop mix.
usage: python tools/isa_mix.py splatco_amd/csrc/blend.hip blend_backward_kernel [extra hipcc flags]"""
import collections
import re
import subprocess
import sys


def classify(op):
    if op.startswith(("v_exp", "v_rcp", "v_log", "v_sqrt", "v_rsq")): return "valu_transcendental"
    if op.startswith("v_permlane"): return "valu_permlane_swap"
    if op.endswith("_dpp") or "_dpp" in op: return "valu_dpp"
    if op.startswith("v_cmp"): return "valu_cmp"
    if op.startswith("v_cndmask"): return "valu_cndmask"
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_"): return "valu_plain"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if op.startswith("s_waitcnt"): return "s_waitcnt"
    if op.startswith("s_nop"): return "s_nop"
    if op.startswith(("s_cbranch", "s_branch", "s_barrier", "s_endpgm")): return "branch_barrier"
    if op.startswith("s_load") or op.startswith("s_buffer_load"): return "smem"
    if op.startswith("s_"): return "salu"
    return "other"


def main(src, kernel, extra):
    asm = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                          "--cuda-device-only", "-S", src, "-o", "-"] + extra, capture_output=True, text=True).stdout
    lines = asm.splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(rf"^_ZN3scr\d+{kernel}\w*:", l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end + 1]
    labels = {}
    insts = []          # (opcode, text)
    for l in body:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        t = l.strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        insts.append((t.split()[0], t))
    total = collections.Counter(classify(op) for op, _ in insts)
    print(f"# {kernel}: {len(insts)} instructions  " + "  ".join(f"{k}={v}" for k, v in sorted(total.items())))
    loops = []
    for i, (op, t) in enumerate(insts):
        if op.startswith("s_cbranch") or op == "s_branch":
            tgt = t.split()[-1]
            if tgt in labels and labels[tgt] <= i:
                loops.append((labels[tgt], i, tgt))
    for a, b, tgt in sorted(loops, key=lambda x: x[1] - x[0]):
        c = collections.Counter(classify(op) for op, _ in insts[a:b + 1])
        print(f"loop {tgt}: instructions {a}..{b} ({b - a + 1})  " + "  ".join(f"{k}={v}" for k, v in sorted(c.items())))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3:])
