#!/usr/bin/env python
"""Reduce the CSVs written by tools/profile_gpu.sh to one JSON: per-kernel mean duration over the
timed region and mean PMC counter values per launch (corrections per MI355X_MICROARCH.md §HBM).

usage: python tools/pmc_summary.py gpurun_out/prof_<tag> [kernel-substring] [last_k] > profiles/...json
"""
import csv
import glob
import json
import os
import sys


def main():
    root = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else "tick_kernel"
    last_k = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    out = {"dir": root, "kernel": pat, "last_k": last_k, "counters": {}}
    # kernel trace
    for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_trace.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if pat in r["Kernel_Name"]]
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows][-last_k:]
        if d:
            out["kernel_us_mean"] = sum(d) / len(d)
            out["kernel_us_min"], out["kernel_us_max"], out["launches"] = min(d), max(d), len(d)
            r = rows[-1]
            out["resources"] = {k: r.get(k) for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Grid_Size", "Workgroup_Size") if k in r}
    for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
        out["stats_csv"] = [r for r in csv.DictReader(open(f))][:8]
    # counters
    for f in sorted(glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
        per = {}
        for r in csv.DictReader(open(f)):
            if pat not in r["Kernel_Name"]:
                continue
            per.setdefault(r["Counter_Name"], {}).setdefault(int(r["Dispatch_Id"]), 0.0)
            per[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
        for name, byd in per.items():
            vals = [byd[k] for k in sorted(byd)][-last_k:]
            out["counters"][name] = sum(vals) / len(vals)
    c = out["counters"]
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        # FETCH_SIZE / WRITE_SIZE are in KiB.  Calibrated on this kernel's access shapes (tools/calib,
        # profiles/r02_hbm_counter_calibration.json): TCC_EA0_RDREQ counts 128-byte requests and FETCH_SIZE prices them
        # at 64 B, so every read pattern is reported at exactly 1/2; WRITE_SIZE is exact.
        out["hbm_read_bytes_raw"] = c["FETCH_SIZE"] * 1024
        out["hbm_read_bytes"] = c["FETCH_SIZE"] * 2048
        out["hbm_write_bytes"] = c["WRITE_SIZE"] * 1024
        out["hbm_bytes_per_launch"] = out["hbm_read_bytes"] + out["hbm_write_bytes"]
        out["calibration"] = "reads = 2 x FETCH_SIZE (measured factor 0.500 for 16-B loads at 16/32/64-B lane stride and 64-B cells), writes = WRITE_SIZE (factor 1.000): profiles/r02_hbm_counter_calibration.json"
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
