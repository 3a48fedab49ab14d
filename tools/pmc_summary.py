#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc CSV output (one directory per pass) into the per-kernel summary committed
under profiles/ and the per-launch HBM traffic table bench.py reads (profiles/hbm_traffic.json).

usage: pmc_summary.py OUT.txt OUT.json DIR [DIR ...]     (each DIR holds *_counter_collection.csv)

Traffic per launch = 2*FETCH_SIZE + WRITE_SIZE (KB): on gfx950 FETCH_SIZE counts 64 B per 128-B request for
wide (16 B/lane) reads (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is taken as is."""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from provenance import stamp


def short(name):
    n = name.split("(")[0].replace("void ", "").replace("scr::", "")
    return n.split("<")[0]


def main(out_txt, out_json, dirs):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for d in dirs:
        for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(path)):
                k = short(row["Kernel_Name"])
                if k.startswith("__amd") or "at::" in row["Kernel_Name"] or "Cijk" in k:
                    continue
                a = acc[k][row["Counter_Name"]]
                a[0] += float(row["Counter_Value"])
                a[1] += 1
    traffic = {}
    with open(out_txt, "w") as f:
        f.write("# rocprofv3 --pmc (separate passes, --kernel-trace only), bench.py cfg1, averages per launch\n")
        f.write("# FETCH_SIZE / WRITE_SIZE in KB; traffic per launch = 2*FETCH_SIZE + WRITE_SIZE (gfx950 correction,\n")
        f.write("# MI355X_MICROARCH.md HBM section); SQ *_CYCLES / WAIT / ACTIVE in quad-cycles, INSTS in instructions\n")
        for k in sorted(acc):
            c = {n: v[0] / v[1] for n, v in acc[k].items()}
            line = f"{k:34s}"
            if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                t = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
                traffic["tile_sort_kernel" if k.startswith("tile_sort") or k.startswith("tile_merge") else k] = \
                    traffic.get("tile_sort_kernel" if k.startswith("tile_sort") or k.startswith("tile_merge") else k, 0) + int(t)
                line += f" FETCH_SIZE {c['FETCH_SIZE']:10.0f} KB  WRITE_SIZE {c['WRITE_SIZE']:10.0f} KB  -> traffic {t / 1e6:8.1f} MB/launch"
            rest = "  ".join(f"{n}={v:.3g}" for n, v in sorted(c.items()) if n not in ("FETCH_SIZE", "WRITE_SIZE"))
            f.write(line + ("  " + rest if rest else "") + "\n")
    insts = {}
    for k in acc:
        if "SQ_INSTS_VALU" in acc[k]:
            key = "tile_sort_kernel" if k.startswith("tile_sort") or k.startswith("tile_merge") else k
            insts[key] = insts.get(key, 0) + int(acc[k]["SQ_INSTS_VALU"][0] / acc[k]["SQ_INSTS_VALU"][1])
    busy = {}
    for k in acc:
        if "SQ_ACTIVE_INST_VALU" in acc[k]:
            key = "tile_sort_kernel" if k.startswith("tile_sort") or k.startswith("tile_merge") else k
            busy[key] = busy.get(key, 0) + int(acc[k]["SQ_ACTIVE_INST_VALU"][0] / acc[k]["SQ_ACTIVE_INST_VALU"][1])
    gui = {}
    for k in acc:
        if "GRBM_GUI_ACTIVE" in acc[k]:    # GPU-active cycles summed over the 8 XCDs: / 8 / kernel time = the clock the kernel ran at
            key = "tile_sort_kernel" if k.startswith("tile_sort") or k.startswith("tile_merge") else k
            gui[key] = gui.get(key, 0) + int(acc[k]["GRBM_GUI_ACTIVE"][0] / acc[k]["GRBM_GUI_ACTIVE"][1])
    prov = stamp("rocprofv3 --kernel-trace --pmc <one counter group per pass> -- python3 bench.py --steps 20 --warmup 5 "
                 "--no-cpu-baseline --no-cfg2 (tools/profile_round.sh); averages per launch")
    for table in (traffic, insts, busy, gui):
        table["_provenance"] = prov
    if len(gui) > 1:
        json.dump(gui, open(out_json.replace("hbm_traffic", "gui_active"), "w"), indent=1, sort_keys=True)
    if len(busy) > 1:   # quad-cycles per launch during which the VALU pipes are busy (bench.py: valu.valu_frac)
        json.dump(busy, open(out_json.replace("hbm_traffic", "valu_busy"), "w"), indent=1, sort_keys=True)
    json.dump(traffic, open(out_json, "w"), indent=1, sort_keys=True)
    if len(insts) > 1:
        json.dump(insts, open(out_json.replace("hbm_traffic", "valu_insts"), "w"), indent=1, sort_keys=True)
    print(open(out_txt).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3:])
