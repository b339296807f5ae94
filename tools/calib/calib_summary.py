#!/usr/bin/env python
"""Divide rocprofv3's FETCH_SIZE / WRITE_SIZE per calibration kernel by the bytes the kernel is known to move
(tools/calib/hbm_calib.hip) -> the factors profiles/r02_pmc_traffic.json applies to the tick kernel's counters.

usage: python tools/calib/calib_summary.py gpurun_out/calib > calibration.json
"""
import csv
import glob
import json
import os
import sys

KERNEL_OF = {"rd16_s16": "rd16(", "rd16_s32": "rd16(", "rd16_s64": "rd16(", "rd64_cell": "rd64_cell(", "rd16_rand": "rd16_rand(",
             "rd4_rand": "rd4_rand(", "wr16_s16": "wr16(", "wr64_quad": "wr64_quad(", "wr16_rand4": "wr16_rand4("}


def main():
    root = sys.argv[1]
    known = json.load(open(os.path.join(root, "known.json")))
    order = [k["name"] for k in known["kernels"]]
    counters = {}  # name -> {counter: value}
    for f in sorted(glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
        rows = list(csv.DictReader(open(f)))
        byd = {}
        for r in rows:
            if "flush_rd" in r["Kernel_Name"] or not any(k.rstrip("(") in r["Kernel_Name"] for k in KERNEL_OF.values()):
                continue  # cache flushes and the runtime's own fill kernels (hipMemset)
            byd.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "c": {}})
            byd[int(r["Dispatch_Id"])]["c"].setdefault(r["Counter_Name"], 0.0)
            byd[int(r["Dispatch_Id"])]["c"][r["Counter_Name"]] += float(r["Counter_Value"])
        seq = [byd[k] for k in sorted(byd)]  # dispatch order == order of known["kernels"] (reps = 1 under the profiler)
        for name, d in zip(order, seq):
            assert KERNEL_OF[name].rstrip("(") in d["name"], (name, d["name"])
            counters.setdefault(name, {}).update(d["c"])
    out = {"region_bytes": known["region_bytes"], "kernels": [],
           "note": "FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them, x1024; factor = counter bytes / known bytes "
                   "(line64 = bytes of the 64-byte lines touched, what has to cross the HBM interface for a sparse pattern)"}
    for k in known["kernels"]:
        c = counters.get(k["name"], {})
        e = dict(k)
        e["counters"] = c
        if "FETCH_SIZE" in c:
            e["fetch_bytes"] = c["FETCH_SIZE"] * 1024
            if k["line64_read_bytes"]:
                e["fetch_over_useful"] = e["fetch_bytes"] / k["useful_read_bytes"]
                e["fetch_over_line64"] = e["fetch_bytes"] / k["line64_read_bytes"]
        if "WRITE_SIZE" in c:
            e["write_size_bytes"] = c["WRITE_SIZE"] * 1024
            if k["write_bytes"]:
                e["write_over_known"] = e["write_size_bytes"] / k["write_bytes"]
        if k["ms"] > 0:
            e["GBps_line64"] = (k["line64_read_bytes"] + k["write_bytes"]) / k["ms"] / 1e6
        out["kernels"].append(e)
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
