#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; counter_collection CSVs) of tools/kprof*.py into HBM
bytes per point per kernel and merge them into profiles/r02_pmc_traffic.json (PMC_OUT overrides the file name).

    python tools/pmc_traffic.py fetch.csv write.csv M      (M = points of the profiled program; the backward
                                                            kernels run on chunks of at most 131072 of them)

Corrections (MI355X_MICROARCH.md, HBM section): both counters are reported in KB; on gfx950 FETCH_SIZE tallies
128-byte requests at 64 bytes, so it is doubled; WRITE_SIZE is taken as is.
"""
import collections
import csv
import json
import os
import sys

NAMES = {"sdf_eval_kernel<2": "sdf_eval_kernel(get_all)", "sdf_eval_kernel<0": "sdf_eval_kernel(forward)",
         "pir::bwd_sweep_kernel": "bwd_sweep_kernel", "wgrad_kernel": "wgrad_kernel", "wgrad256_kernel": "wgrad256_kernel",
         "mlp_eval_kernel": "mlp_eval_kernel", "mlp_bwd_sweep_kernel": "mlp_bwd_sweep_kernel"}
SUFFIX = os.environ.get("PMC_SUFFIX", "")     # e.g. "(diffuse)": one entry per profiled network


def per_launch(path, counter):
    tot, n = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        for frag, name in NAMES.items():
            if frag in r["Kernel_Name"]:
                tot[name] += float(r["Counter_Value"]) * 1024.0
                n[name] += 1
                break
    return {k: tot[k] / n[k] for k in tot}, n


def main():
    fetch, nf = per_launch(sys.argv[1], "FETCH_SIZE")
    write, _ = per_launch(sys.argv[2], "WRITE_SIZE")
    pts = float(sys.argv[3])
    out_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", os.environ.get("PMC_OUT", "r02_pmc_traffic.json"))
    out = json.load(open(out_path)) if os.path.exists(out_path) else {}
    for k in fetch:
        if SUFFIX and not k.startswith("mlp_"):
            continue
        n = min(pts, float(os.environ.get("PMC_CHUNK", "262144"))) if k in ("bwd_sweep_kernel", "wgrad_kernel", "wgrad256_kernel") else pts
        out[k + SUFFIX] = {"points_per_launch": n, "fetch_bytes_corrected": 2.0 * fetch[k], "write_bytes": write.get(k, 0.0),
                  "hbm_bytes_per_point": (2.0 * fetch[k] + write.get(k, 0.0)) / n}
        print(k + SUFFIX, out[k + SUFFIX], "dispatches", nf[k])
    json.dump(out, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
