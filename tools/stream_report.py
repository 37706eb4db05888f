#!/usr/bin/env python3
"""profiles/rNN_stream_kernels.json from the three rocprofv3 passes of tools/kprof_stream.py:
    python tools/stream_report.py STATS_DIR FETCH_DIR WRITE_DIR P OUT.json
Per kernel: average duration (rocprofv3 --kernel-trace --stats), algorithmic bytes per unit and the
achieved rate, HBM traffic from the PMC passes (FETCH_SIZE doubled on gfx950, WRITE_SIZE as is, KB units:
MI355X_MICROARCH.md), fractions of the 8 TB/s peak and of the 6.29 TB/s a float4 copy reaches."""
import collections
import csv
import glob
import json
import os
import sys

# kernel-name fragment -> (label, algorithmic bytes per unit, unit)
KERNELS = [
    ("ggx_forward", "ggx_forward_kernel", 104, "point"),
    ("ggx_backward", "ggx_backward_kernel", 172, "point"),
    ("vis_points_kernel", "vis_points_kernel", 12, "sample"),
    ("vis_composite_kernel", None, None, "sample"),        # forward 16 B / backward 32 B: split by template argument
    ("shade_inputs_forward", "shade_inputs_forward_kernel", 64, "point"),
    ("shade_inputs_backward", "shade_inputs_backward_kernel", 124, "point"),
    ("embed_out_kernel", "embed_out_kernel (L = 10)", 12 + 63 * 4, "point"),
    ("embed_in_kernel", "embed_in_kernel (L = 10)", 12 + 63 * 4 + 12, "point"),
]


def label(name):
    for frag, lab, by, unit in KERNELS:
        if frag in name:
            if lab is None:
                bwd = "true" in name.split("vis_composite_kernel")[1][:40] or ", 1>" in name or ",1>" in name
                return ("vis_composite_kernel(backward)" if bwd else "vis_composite_kernel(forward)"), (32 if bwd else 16), unit
            return lab, by, unit
    return None


def read_csv(d, pattern):
    rows = []
    for f in glob.glob(os.path.join(d, "**", pattern), recursive=True):
        rows += list(csv.DictReader(open(f)))
    return rows


def main():
    stats_dir, fetch_dir, write_dir, P, out_path = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5]
    n_samples = (P // 118) * 118
    dur = collections.defaultdict(list)
    for r in read_csv(stats_dir, "*kernel_trace.csv"):
        lab = label(r["Kernel_Name"])
        if lab:
            dur[lab].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    pmc = {"FETCH_SIZE": collections.defaultdict(list), "WRITE_SIZE": collections.defaultdict(list)}
    for d, ctr in ((fetch_dir, "FETCH_SIZE"), (write_dir, "WRITE_SIZE")):
        for r in read_csv(d, "*counter_collection.csv"):
            if r["Counter_Name"] == ctr:
                lab = label(r["Kernel_Name"])
                if lab:
                    pmc[ctr][lab].append(float(r["Counter_Value"]) * 1024.0)
    sq = collections.defaultdict(lambda: collections.defaultdict(list))
    if len(sys.argv) > 6:
        for r in read_csv(sys.argv[6], "*counter_collection.csv"):
            lab = label(r["Kernel_Name"])
            if lab:
                sq[lab][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for lab, ds in dur.items():
        name, by, unit = lab
        units = P if unit == "point" else n_samples
        ds = sorted(ds)[: max(1, len(ds))]
        avg_ns = sum(ds[1:]) / max(1, len(ds) - 1) if len(ds) > 1 else ds[0]     # drop the fastest? no: drop nothing but the first sorted
        avg_ns = sum(ds) / len(ds)
        alg = by * units
        row = {"units": units, "unit": unit, "launches": len(ds), "avg_us": round(avg_ns / 1e3, 2),
               "algorithmic_bytes_per_unit": by, "achieved_GBps": round(alg / avg_ns, 1),
               "frac_of_8TBps_peak": round(alg / avg_ns / 8000.0, 4), "frac_of_6.29TBps_copy": round(alg / avg_ns / 6290.0, 4)}
        f, w = pmc["FETCH_SIZE"].get(lab), pmc["WRITE_SIZE"].get(lab)
        if f and w:
            traffic = 2.0 * sum(f) / len(f) + sum(w) / len(w)
            row.update({"hbm_fetch_bytes_corrected": round(2.0 * sum(f) / len(f)), "hbm_write_bytes": round(sum(w) / len(w)),
                        "hbm_bytes_per_unit": round(traffic / units, 1), "traffic_over_algorithmic": round(traffic / alg, 3),
                        "hbm_GBps_by_pmc": round(traffic / avg_ns, 1)})
        c = {k: sum(v) / len(v) for k, v in sq.get(lab, {}).items()}
        if c.get("GRBM_GUI_ACTIVE") and c.get("SQ_WAVE_CYCLES"):
            # gfx9 VALUBusy: vector-ALU issue cycles (counted in quads) per SIMD cycle; GRBM_GUI_ACTIVE sums the 8 XCDs
            xcd_cycles = c["GRBM_GUI_ACTIVE"] / 8.0
            row.update({"valu_busy_share_of_simd_cycles": round(c.get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0 / 1024.0 / xcd_cycles, 4),
                        "valu_insts_per_unit": round(c.get("SQ_INSTS_VALU", 0.0) * 64.0 / units, 1) if c.get("SQ_INSTS_VALU") else None,
                        "wave_wait_any_share": round(c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"], 4),
                        "wave_issue_stall_share": round(c.get("SQ_WAIT_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"], 4),
                        "wave_active_share": round(c.get("SQ_ACTIVE_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"], 4)})
        out[name] = row
        print(name, row)
    json.dump(out, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
