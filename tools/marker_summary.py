#!/usr/bin/env python3
"""Per-stage summary of a `rocprofv3 --marker-trace --kernel-trace --output-format csv` run of this library with its stage
markers on (SPLATCO_MARKERS=1): the LAST training step's roctx ranges as a tree (host time of each range, which in marker
mode closes behind a device synchronisation for the host stages), and under every host stage the device time of the
kernels that ran inside it.
usage: marker_summary.py DIR OUT.txt "command line"  """
import collections
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from provenance import stamp

HOST_STAGES = ("prefilter_voxel", "render", "generate_neural_gaussians", "rasterize", "loss", "consistency_loss", "backward",
               "gradient_exchange", "tv_loss", "training_statis", "optimizer_step")


def col(row, *names):
    for n in names:
        if n in row and row[n] != "":
            return row[n]
    raise KeyError(names)


def main(d, out_txt, command):
    marks, kerns = [], []
    for path in glob.glob(d + "/**/*marker_api_trace.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            marks.append((int(col(r, "Start_Timestamp")), int(col(r, "End_Timestamp")), col(r, "Function", "Message", "Name")))
    for path in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            kerns.append((int(col(r, "Start_Timestamp")), int(col(r, "End_Timestamp")), col(r, "Kernel_Name")))
    marks.sort()
    kerns.sort()
    if not marks:
        open(out_txt, "w").write("# no roctx ranges in the trace (markers off?)\n")
        return
    # the last step = from the last "prefilter_voxel" range of the first view of a step; take everything after the last
    # optimizer_step-before-last, i.e. the final occurrence window [start of the last step's first stage, end of trace]
    opt = [m for m in marks if m[2] == "optimizer_step"]
    t_lo = opt[-2][1] if len(opt) >= 2 else marks[0][0]
    t_hi = opt[-1][1] if opt else marks[-1][1]
    step = [m for m in marks if m[0] >= t_lo and m[1] <= t_hi]
    ks = [k for k in kerns if k[0] >= t_lo and k[1] <= t_hi]
    short = lambda n: n.split("(")[0].replace("void ", "").replace("scr::", "").split("<")[0]
    with open(out_txt, "w") as f:
        prov = stamp(command)
        f.write(f"# {command}\n# collected at git {prov.get('git')} on {prov.get('date')}, library {prov.get('library_sha256_16')}\n")
        f.write("# ONE training step (the last one of the run), roctx ranges nested by time; host stages close behind a device\n"
                "# synchronisation in marker mode, so a stage's wall time covers its kernels.  ms(host) = the range on the host\n"
                "# time line; ms(device) = sum of the durations of the kernels that ran inside it; kernels = their count.\n")
        f.write(f"# step: {(t_hi - t_lo) / 1e6:.3f} ms wall, {len(ks)} kernels, {sum(k[1] - k[0] for k in ks) / 1e6:.3f} ms of kernel time\n\n")
        # nesting depth by containment
        stack = []
        for (a, b, name) in step:
            while stack and a >= stack[-1][1]:
                stack.pop()
            depth = len(stack)
            stack.append((a, b))
            inside = [k for k in ks if k[0] >= a and k[1] <= b] if name in HOST_STAGES else []
            line = f"{'  ' * depth}{name:<{44 - 2 * depth}s} {((b - a) / 1e6):9.3f} ms(host)"
            if inside:
                line += f" {sum(k[1] - k[0] for k in inside) / 1e6:9.3f} ms(device) {len(inside):5d} kernels"
            f.write(line + "\n")
        f.write("\n# kernels of the step by host stage (innermost host stage that contains them), device ms\n")
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        host = [(a, b, n) for (a, b, n) in step if n in HOST_STAGES]
        for (ka, kb, kn) in ks:
            owner = "(outside every stage)"
            for (a, b, n) in host:      # sorted by start: the last containing one is the innermost
                if ka >= a and kb <= b:
                    owner = n
            per[owner][short(kn)] += (kb - ka) / 1e6
        for owner in sorted(per, key=lambda o: -sum(per[o].values())):
            f.write(f"{owner:<28s} {sum(per[owner].values()):9.3f} ms\n")
            for kn, ms in sorted(per[owner].items(), key=lambda kv: -kv[1])[:12]:
                f.write(f"    {kn:<56s} {ms:9.3f}\n")


if __name__ == "__main__":
    main(*sys.argv[1:4])
