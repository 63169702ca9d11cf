#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (separate runs under DIR/*/...counter_collection.csv) per kernel.

  pmc_summarize.py DIR [--evals-per-launch N] [--kernel-ms T]  > profiles/<round>_pmc_summary.json

Counter values are summed over the counter's instances within a dispatch (rocprofv3 emits one row per
dispatch/counter after aggregation) and averaged over dispatches. Derived figures follow
MI355X_MICROARCH.md: HBM traffic = 2 * FETCH_SIZE (gfx950 half-count correction) + WRITE_SIZE, both in KB.
"""
import argparse, csv, glob, json, os
from collections import defaultdict

ap = argparse.ArgumentParser()
ap.add_argument("dir")
ap.add_argument("--evals-per-launch", type=float, default=0)
ap.add_argument("--kernel-ms", type=float, default=0)
ap.add_argument("--workload", default="")
ap.add_argument("--code", default="", help="config.code of the bench line: the key of the kernels that were profiled")
ap.add_argument("--clock-ghz", type=float, default=2.1, help="shader clock under VALU-dense load (class_rate prints it per class)")
ap.add_argument("--command", default="python bench.py --steps 5 --warmup 1 --no-cpu-baseline")
a = ap.parse_args()

KERNELS = ("leaf_eval_kernel", "leaf_dense_kernel", "march_records_kernel", "march_dense_kernel", "leaf_kernel", "prune_kernel", "prune_spec_kernel", "prune_resolve_kernel", "dc_block_test_kernel", "dc_grid_clear_kernel", "dc_origin_kernel", "dc_edges_kernel", "dc_normals_kernel", "dc_place_kernel", "dc_quads_kernel", "stl_kernel", "leaf_brick_kernel", "eval_kernel", "flat_grid_kernel", "flat_march_kernel", "flat_cut_scan_kernel", "flat_march_list_kernel")
acc = {k: defaultdict(lambda: [0.0, 0]) for k in KERNELS}
for f in glob.glob(os.path.join(a.dir, "**", "*counter_collection.csv"), recursive=True):
    per = defaultdict(float)
    for r in csv.DictReader(open(f)):
        for k in KERNELS:
            if r["Kernel_Name"].startswith(k) or (" " + k) in r["Kernel_Name"]:
                per[(k, r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
                break
    for (k, _, name), v in per.items():
        acc[k][name][0] += v
        acc[k][name][1] += 1
out = {"command": "rocprofv3 --kernel-trace --pmc <counters> --output-format csv -- " + a.command,
       "note": "separate --pmc passes; values are averages per dispatch", "workload": a.workload, "code": a.code}
for k in KERNELS:
    if not acc[k]:
        continue
    d = {n: v[0] / v[1] for n, v in sorted(acc[k].items())}
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        d["hbm_traffic_gb_per_launch"] = (2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024 / 1e9
    if "SQ_INSTS_VALU" in d:
        if "SQ_INSTS_SALU" in d:
            d["salu_per_valu"] = d["SQ_INSTS_SALU"] / d["SQ_INSTS_VALU"]
        if k in ("leaf_kernel", "leaf_eval_kernel") and a.evals_per_launch:
            d["valu_lane_instr_per_eval"] = d["SQ_INSTS_VALU"] * 64 / a.evals_per_launch
            if a.kernel_ms:
                rate = d["SQ_INSTS_VALU"] * 64 / (a.kernel_ms * 1e-3)
                d["valu_lane_instr_per_s"] = rate
                # 256 CUs x 4 SIMDs x 32 lanes/clk x 2.4 GHz (MI355X_MICROARCH.md: SIMD-32, 2 cycles per wave64 VALU op)
                d["valu_issue_frac_of_peak"] = rate / (256 * 4 * 32 * 2.4e9)
    # What the kernel's instruction MIX allows: VALU issue rates differ by class on gfx950 (tools/ubench/class_rate.hip, profiles/
    # r5_valu_class_rates.txt: REAL shader-clock cycles from s_memtime at 4 waves per SIMD, and the clock the loops ran at): f32 add /
    # mul / fma with register, literal or inline operands 2.44-2.52 cycles per wave-instruction and SIMD; min / max / med3, compares,
    # selects, shifts, conversions, DPP, anything with an SGPR operand 4.1-4.4; f64 4.3; rcp / sqrt 8.2 -- at 1.9-2.4 GHz (VALU-dense
    # loops do not hold the 2.4 GHz peak clock: fma 1.93, add / mul 2.0-2.1, the 4-cycle classes 2.2-2.4). Lower bound of the
    # kernel's duration = sum over classes of count x cycles / (1024 SIMDs x clock), the unclassified instructions (moves, logic,
    # selects, compares, min / max) priced at the 4.25 of the majority among them, the clock at --clock-ghz (2.1: the mix's own).
    need = ("SQ_INSTS_VALU", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32")
    if all(n in d for n in need):
        fast = d["SQ_INSTS_VALU_ADD_F32"] + d["SQ_INSTS_VALU_MUL_F32"] + d["SQ_INSTS_VALU_FMA_F32"]
        f64 = sum(d.get(n, 0.0) for n in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64"))
        trans = d["SQ_INSTS_VALU_TRANS_F32"]
        other = max(0.0, d["SQ_INSTS_VALU"] - fast - f64 - trans)
        cyc = fast * 2.47 + f64 * 4.3 + trans * 8.2 + other * 4.25
        d["valu_mix"] = {"f32_add_mul_fma": fast / d["SQ_INSTS_VALU"], "f64": f64 / d["SQ_INSTS_VALU"], "trans": trans / d["SQ_INSTS_VALU"],
                         "other (min/max, compare, select, move, logic, int, cvt, dpp)": other / d["SQ_INSTS_VALU"],
                         "cycles_per_instr_by_class": cyc / d["SQ_INSTS_VALU"], "clock_ghz": a.clock_ghz, "roof_ms": cyc / (1024 * a.clock_ghz * 1e9) * 1e3}
        if a.kernel_ms:
            d["valu_mix"]["kernel_ms"] = a.kernel_ms
            d["valu_mix"]["frac_of_mix_roof"] = d["valu_mix"]["roof_ms"] / a.kernel_ms
    if "SQ_THREAD_CYCLES_VALU" in d and d.get("SQ_ACTIVE_INST_VALU"):
        d["valu_active_lanes_of_64"] = d["SQ_THREAD_CYCLES_VALU"] / d["SQ_ACTIVE_INST_VALU"]
    if "GRBM_GUI_ACTIVE" in d:
        d["clock_ghz_est"] = None
    if "SQ_WAIT_ANY" in d and "SQ_WAVE_CYCLES" in d:
        d["wave_wait_any_frac"] = d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"]
    if "SQ_WAIT_INST_ANY" in d and "SQ_WAVE_CYCLES" in d:
        d["wave_wait_inst_frac"] = d["SQ_WAIT_INST_ANY"] / d["SQ_WAVE_CYCLES"]
    out[k] = d
print(json.dumps(out, indent=1))
