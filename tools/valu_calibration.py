#!/usr/bin/env python3
"""Join the VALU calibration probe's own output (tools/exp/valu_calib: wall time, shader clock, cycles per instruction
per SIMD) with the rocprofv3 --pmc readings of the same launches, per instruction form.

usage: valu_calibration.py OUT.txt OUT.json BARE.jsonl PMC_DIR [PMC_DIR ...]

OUT.json is what bench.py normalises `roofline.valu` with:
  busy_quads_per_cycle_per_simd_saturated  SQ_ACTIVE_INST_VALU per (SIMD x shader cycle) of the pure v_fma_f32 probe at
                                           its saturating occupancy -- the counter's rate when the vector pipe never idles
  counter_per_inst[form]                   SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU
  cycles_per_inst[form]                    measured issue cost per wave64 instruction per SIMD (shader cycles)
  shader_clock_MHz                         s_memtime ticks per 100 MHz s_memrealtime tick during the probes
"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from provenance import stamp

SIMDS = 1024


def main(out_txt, out_json, bare, dirs):
    rows = [json.loads(l) for l in open(bare) if l.startswith("{")]
    pmc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(path)):
                name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
                pmc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {"forms": {}, "note": "every probe launch runs twice (10 warm-up iterations, then 4000): the larger reading is the timed launch"}
    lines = ["# VALU calibration on MI355X (tools/exp/valu_calib.hip; tools/profile_valu_calib.sh)",
             "# cyc/inst = shader cycles per wave64 instruction per SIMD at the MEASURED clock; ACTIVE/INST = SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU;",
             "# busy = SQ_ACTIVE_INST_VALU / (1024 SIMDs x kernel shader cycles) -- the counter's own 'fraction busy' in its native unit",
             f"{'form':58s} {'w/SIMD':>6s} {'ms':>8s} {'MHz':>7s} {'cyc/inst':>8s} {'INSTS_VALU':>12s} {'ACTIVE_VALU':>12s} {'ACTIVE/INST':>11s} {'busy':>6s} {'GUI_ACTIVE MHz':>14s}"]
    for r in rows:
        c = {k: max(v) for k, v in pmc.get(r["kernel"], {}).items()}
        insts, act = c.get("SQ_INSTS_VALU"), c.get("SQ_ACTIVE_INST_VALU")
        cycles = r["ms"] * 1e-3 * r["shader_clock_MHz"] * 1e6
        busy = act / (SIMDS * cycles) if act else None
        gui = c.get("GRBM_GUI_ACTIVE")
        entry = dict(r, SQ_INSTS_VALU=insts, SQ_ACTIVE_INST_VALU=act, SQ_WAVE_CYCLES=c.get("SQ_WAVE_CYCLES"),
                     SQ_BUSY_CYCLES=c.get("SQ_BUSY_CYCLES"), GRBM_GUI_ACTIVE=gui, SQ_ACTIVE_INST_ANY=c.get("SQ_ACTIVE_INST_ANY"),
                     active_per_inst=(act / insts if act and insts else None), busy_quads_per_cycle_per_simd=busy)
        out["forms"][f"{r['op']} @{r['waves_per_simd']}"] = entry
        lines.append(f"{r['op'][:58]:58s} {r['waves_per_simd']:6d} {r['ms']:8.3f} {r['shader_clock_MHz']:7.0f} {r['cycles_per_inst_per_simd']:8.2f} "
                     f"{insts or 0:12.4g} {act or 0:12.4g} {(act / insts) if act and insts else 0:11.3f} {busy or 0:6.3f} "
                     f"{(gui / (r['ms'] * 1e-3) / 1e6) if gui else 0:14.0f}")
    sat = [e for k, e in out["forms"].items() if k.startswith("v_fma_f32") and e["busy_quads_per_cycle_per_simd"]]
    if sat:
        best = max(sat, key=lambda e: e["busy_quads_per_cycle_per_simd"])
        out["busy_quads_per_cycle_per_simd_saturated"] = best["busy_quads_per_cycle_per_simd"]
        out["saturating_probe"] = f"v_fma_f32 @{best['waves_per_simd']} waves/SIMD: {best['cycles_per_inst_per_simd']:.2f} cycles per instruction"
        lines.append(f"# saturated vector pipe (pure v_fma_f32, {best['waves_per_simd']} waves/SIMD): SQ_ACTIVE_INST_VALU = "
                     f"{best['busy_quads_per_cycle_per_simd']:.4f} per SIMD per shader cycle -> valu_frac(kernel) = "
                     f"SQ_ACTIVE_INST_VALU / (1024 x cycles x {best['busy_quads_per_cycle_per_simd']:.4f})")
    out["shader_clock_MHz"] = sum(r["shader_clock_MHz"] for r in rows) / max(len(rows), 1)
    out["cycles_per_inst"] = {k: e["cycles_per_inst_per_simd"] for k, e in out["forms"].items()}
    out["counter_per_inst"] = {k: e["active_per_inst"] for k, e in out["forms"].items()}
    out["_provenance"] = stamp("tools/profile_valu_calib.sh: tools/exp/valu_calib bare + 2 rocprofv3 --pmc passes")
    open(out_txt, "w").write("\n".join(lines) + "\n")
    json.dump(out, open(out_json, "w"), indent=1, sort_keys=True)
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4:])
