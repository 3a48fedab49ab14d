#!/usr/bin/env python3
"""HBM traffic per step of an anchor configuration (bench.py --config cfg2..4) from two rocprofv3 --pmc passes
(FETCH_SIZE, WRITE_SIZE; --kernel-trace only), by kernel and by bench.py's kernel classes.

usage: pmc_cfg_summary.py OUT.txt OUT.json DIR_FETCH DIR_WRITE

Traffic = 2*FETCH_SIZE + WRITE_SIZE (KB; the gfx950 correction of MI355X_MICROARCH.md for wide reads, an upper bound for
narrow gathers).  Steps = launches of filter_kernel (one per step)."""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from provenance import stamp

CLASSES = (  # kernel-name prefix -> bench.py kernel class (capi.hip kProfNames)
    ("filter_kernel", "filter_kernel"), ("preprocess_backward", "preprocess_backward_kernel"),
    ("preprocess_kernel", "preprocess_kernel"), ("plan_scan", "plan_scan_kernel"), ("scatter_kernel", "scatter_kernel"),
    ("tile_sort", "tile_sort_kernel"), ("tile_merge", "tile_sort_kernel"), ("tile_order", "blend_forward_kernel"),
    ("blend_forward", "blend_forward_kernel"), ("blend_backward", "blend_backward_kernel"),
    ("expand_backward", "expand_backward_kernel"), ("expand_", "expand_kernel"),
    ("triplane_forward", "triplane_forward_kernel"), ("plane_row_pairs", "triplane_forward_kernel"),
    ("tp_", "plane_sample_backward_kernels"), ("mlp_heads_forward", "mlp_heads_kernel"),
    ("mlp_heads_", "mlp_heads_backward_kernel"), ("nl_bwd", "norm_linear_backward_kernels"), ("nl_", "norm_linear_kernels"),
    ("tpa_", "plane_attention_kernels"), ("l1_ssim_forward", "l1_ssim_forward_kernel"),
    ("l1_ssim_backward", "l1_ssim_backward_kernel"), ("anchor_gather_backward", "anchor_gather_backward_kernel"),
    ("anchor_gather", "anchor_gather_kernel"),
)


def short(name):
    return name.split("(")[0].replace("void ", "").replace("scr::", "").split("<")[0]


def read(d, counter):
    tot, n = collections.Counter(), collections.Counter()
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(path)):
            if row["Counter_Name"] != counter or "scr::" not in row["Kernel_Name"]:
                continue
            k = short(row["Kernel_Name"])
            tot[k] += float(row["Counter_Value"])
            n[k] += 1
    return tot, n


def main(out_txt, out_json, d_fetch, d_write):
    fetch, nf = read(d_fetch, "FETCH_SIZE")
    write, nw = read(d_write, "WRITE_SIZE")
    # the two passes are two runs of bench.py, and its settle phase may take a different number of steps in each: every
    # pass is normalised by ITS OWN step count (round 5: the write pass of r05z ran 9 steps, the fetch pass 7, and dividing
    # both by 7 inflated every WRITE_SIZE by 9 / 7)
    steps = max(nf.get("filter_kernel", 1), 1)
    steps_w = max(nw.get("filter_kernel", 1), 1)
    by_class = collections.Counter()
    with open(out_txt, "w") as f:
        f.write("# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only), per STEP of the configuration\n")
        f.write(f"# ({steps} steps in the FETCH_SIZE pass, {steps_w} in the WRITE_SIZE pass); traffic = 2*FETCH_SIZE + WRITE_SIZE (gfx950 correction, MI355X_MICROARCH.md HBM section)\n")
        f.write(f"{'kernel':34s} {'launches/step':>13s} {'fetch MB':>10s} {'write MB':>10s} {'traffic MB':>11s}  class\n")
        for k in sorted(fetch, key=lambda k: -(2 * fetch[k] + write.get(k, 0))):
            t = 2 * fetch[k] * 1024 / steps + write.get(k, 0) * 1024 / steps_w
            cls = next((c for p, c in CLASSES if k.startswith(p)), k)
            by_class[cls] += t
            f.write(f"{k:34s} {nf[k] / steps:13.1f} {fetch[k] * 1024 / steps / 1e6:10.1f} {write.get(k, 0) * 1024 / steps_w / 1e6:10.1f} "
                    f"{t / 1e6:11.1f}  {cls}\n")
    table = {k: int(v) for k, v in by_class.items()}
    table["_provenance"] = stamp("rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE -- python3 bench.py --config <cfg> --steps 3 "
                                 "--warmup 2 --no-cpu-baseline (tools/profile_cfg_pmc.sh); bytes per step by kernel class")
    json.dump(table, open(out_json, "w"), indent=1, sort_keys=True)
    print(open(out_txt).read())


if __name__ == "__main__":
    main(*sys.argv[1:5])
