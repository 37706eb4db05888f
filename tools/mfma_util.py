#!/usr/bin/env python3
"""MFMA utilisation and effective clock per kernel from one rocprofv3 --pmc pass of tools/kprof_mfma.py
(tools/profile_mfma.sh):

    python tools/mfma_util.py <counter_collection.csv> [out.json]

Per dispatch (MI355X_MICROARCH.md, 'DVFS give-back' and the PMC-slot table):
  clock      = GRBM_GUI_ACTIVE / 8 / wall time         (rocprofv3 sums the counter over the 8 XCDs)
  mfma_util  = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs)
               = share of SIMD-cycles at the clock actually held in which the matrix pipe was busy
  SQ_WAVE_CYCLES / SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY: where a wave's lifetime goes
  (parked on s_waitcnt or a barrier / stalled at issue / issuing); the three are disjoint shares of WAVE_CYCLES.
The first third of each kernel's dispatches is dropped (clock still ramping); the median of the rest is reported.
"""
import collections
import csv
import json
import statistics
import sys

KERNELS = {"sdf_eval_kernel<2": "sdf_eval_kernel(get_all)", "pir::bwd_sweep_kernel": "bwd_sweep_kernel",
           "pir::wgrad256_kernel": "wgrad256_kernel"}
# v_mfma_f32_32x32x2_f32: 4096 FLOP and 64 busy cycles on one SIMD per instruction
FLOP_PER_BUSY_CYCLE = 4096.0 / 64.0
N_SIMD = 256 * 4


def main():
    disp = collections.defaultdict(dict)
    for r in csv.DictReader(open(sys.argv[1])):
        for frag, name in KERNELS.items():
            if frag in r["Kernel_Name"]:
                d = disp[(name, int(r["Dispatch_Id"]))]
                d[r["Counter_Name"]] = float(r["Counter_Value"])
                d["ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                break
    per = collections.defaultdict(list)
    for (name, did), d in sorted(disp.items()):
        per[name].append(d)
    out = {}
    for name, ds in per.items():
        ds = ds[len(ds) // 3:]
        med = lambda f: statistics.median(f(d) for d in ds)
        clk = med(lambda d: d["GRBM_GUI_ACTIVE"] / 8.0 / d["ns"])                   # GHz
        util = med(lambda d: d["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["GRBM_GUI_ACTIVE"] / 8.0 * N_SIMD))
        tfl = med(lambda d: d["SQ_VALU_MFMA_BUSY_CYCLES"] * FLOP_PER_BUSY_CYCLE / d["ns"] / 1e3)
        wc = med(lambda d: d["SQ_WAVE_CYCLES"])
        out[name] = {
            "dispatches_used": len(ds), "median_ms": round(med(lambda d: d["ns"]) / 1e6, 3),
            "effective_clock_GHz": round(clk, 3),
            "mfma_busy_share_of_simd_cycles": round(util, 4),
            "executed_mfma_TFLOPs": round(tfl, 1),
            "peak_at_effective_clock_TFLOPs": round(N_SIMD * FLOP_PER_BUSY_CYCLE * clk / 1e3, 1),
            "wave_cycles_share": {"wait_any": round(med(lambda d: d["SQ_WAIT_ANY"]) / wc, 3),
                                  "wait_inst_any": round(med(lambda d: d["SQ_WAIT_INST_ANY"]) / wc, 3),
                                  "active_inst_any": round(med(lambda d: d["SQ_ACTIVE_INST_ANY"]) / wc, 3)},
            "busy_cu_cycles_per_gui_cycle": round(med(lambda d: d["SQ_BUSY_CU_CYCLES"] / (d["GRBM_GUI_ACTIVE"] / 8.0)), 1),
        }
        print(name, json.dumps(out[name]))
    if len(sys.argv) > 2:
        json.dump(out, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
