#!/usr/bin/env python3
"""Reduce the rocprofv3 PMC passes of tools/profile_round.sh to profiles/<round>[_configN]_pmc_summary.json.

    python tools/pmc_summary.py <dir> <out.json> [prefix]

Input: <dir>/<prefix>{fetch,write,mfma}_counter_collection.csv -- three SEPARATE --pmc passes of the same command
(FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE), as MI355X_MICROARCH.md prescribes.
Per ggnn kernel (averaged over its launches):
  hbm_bytes_fetch_x2_plus_write = 2 * FETCH_SIZE_KB * 1024 + WRITE_SIZE_KB * 1024     (gfx950: FETCH_SIZE is doubled)
  gpu_cycles = GRBM_GUI_ACTIVE / 8 XCDs;  mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (gpu_cycles * 1024 SIMDs)
  avg_us from the dispatch timestamps of the mfma pass (counter collection serialises kernels).
"_meta" records the sha1 of the kernel sources the passes were taken with: bench.py attaches `traffic` only when it
matches the tree it runs from.
"""
import csv
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def short_name(kernel_name):
    m = re.search(r"ggnn::(\w+?)(?:_kernel)?(<[^(]*>)?\(", kernel_name) or re.search(r"(\w+?)(?:_kernel)?(<[^(]*>)?\(", kernel_name)
    if not m:
        return kernel_name
    base, targs = m.group(1), m.group(2) or ""
    base = base[5:] if base.startswith("ggnn_") else base
    return base + (targs if base.startswith(("gru_fused", "gru_panel", "gemm", "msg_transform_compact")) else "")


def read_pass(path):
    per = defaultdict(lambda: defaultdict(list))      # kernel -> counter -> per-dispatch values
    dur = defaultdict(dict)                           # kernel -> dispatch -> ns
    if not os.path.exists(path):
        return per, dur
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"]
            if "ggnn" not in name:
                continue
            k = short_name(name)
            per[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
            dur[k][row["Dispatch_Id"]] = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
    return per, dur


def main():
    src, dst = sys.argv[1], sys.argv[2]
    prefix = sys.argv[3] if len(sys.argv) > 3 else ""
    fetch, _ = read_pass(os.path.join(src, prefix + "fetch_counter_collection.csv"))
    write, _ = read_pass(os.path.join(src, prefix + "write_counter_collection.csv"))
    mfma, dur = read_pass(os.path.join(src, prefix + "mfma_counter_collection.csv"))
    mean = lambda xs: sum(xs) / len(xs) if xs else 0.0
    import bench
    out = {"_meta": {"csrc_sha1": bench.csrc_sha1(), "prefix": prefix,
                     "counters": "FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE (three separate rocprofv3 --pmc passes)"}}
    for k in sorted(mfma):
        fkb, wkb = mean(fetch[k]["FETCH_SIZE"]), mean(write[k]["WRITE_SIZE"])
        cyc = mean(mfma[k]["GRBM_GUI_ACTIVE"]) / 8.0
        busy = mean(mfma[k]["SQ_VALU_MFMA_BUSY_CYCLES"])
        us = mean(list(dur[k].values())) / 1e3
        out[k] = {"FETCH_SIZE_KB": fkb, "WRITE_SIZE_KB": wkb, "hbm_bytes_fetch_x2_plus_write": 2 * fkb * 1024 + wkb * 1024,
                  "avg_us": us, "launches": len(dur[k]), "mfma_busy_cycles": busy, "gpu_cycles": cyc,
                  "clock_GHz": cyc / us / 1e3 if us else 0.0, "mfma_util": busy / (cyc * 1024) if cyc else 0.0}
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    for k, v in out.items():
        if k != "_meta":
            print("%-44s %8.1f us  hbm %7.1f MB  mfma_util %.3f  clk %.2f GHz" % (k, v["avg_us"], v["hbm_bytes_fetch_x2_plus_write"] / 1e6,
                                                                                  v["mfma_util"], v["clock_GHz"]))


if __name__ == "__main__":
    main()
