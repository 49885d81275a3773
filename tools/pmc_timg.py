#!/usr/bin/env python3
"""Counter sums of two imagination-only runs (R1, R2 replays) -> MFMA utilisation of ONE rollout (tools/pmc_timg.sh).

    python tools/pmc_timg.py counters_R1.csv counters_R2.csv run_R2.json out.json

per rollout: busy = d(SQ_VALU_MFMA_BUSY_CYCLES) / dR   (matrix-pipe busy cycles summed over the 1024 SIMDs)
             active = d(GRBM_GUI_ACTIVE) / 8 / dR       (cycles a dispatch was active, summed over the rollout's dispatches)
  mfma_busy_over_kernel_time = busy / (active * 1024)           (launch gaps excluded)
  mfma_busy_over_T_img       = busy / (T_img * f_clk * 1024)    (wall time of the replay; f_clk = active cycles / summed
                                                                 kernel time is not available here, 2.4 GHz nominal)
  mfma_flops                 = d(SQ_INSTS_VALU_MFMA_MOPS_F32) * 512 / dR  (what the MFMA kernels execute)
"""
import collections
import csv
import json
import re
import sys


def sums(path):
    tot = collections.defaultdict(float)
    per_kernel = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(int)
    for r in csv.DictReader(open(path)):
        tot[r["Counter_Name"]] += float(r["Counter_Value"])
        per_kernel[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            n[r["Kernel_Name"]] += 1
    return tot, per_kernel, n


def main():
    c1, c2, run2, out = sys.argv[1:5]
    t1, k1, n1 = sums(c1)
    t2, k2, n2 = sums(c2)
    r1, r2 = [int(re.search(r"pmc_(\d+)", p).group(1)) for p in (c1, c2)]
    dr = r2 - r1
    t_img_ms = json.loads([ln for ln in open(run2) if ln.startswith("{")][-1])["T_img_ms"]
    busy = (t2["SQ_VALU_MFMA_BUSY_CYCLES"] - t1["SQ_VALU_MFMA_BUSY_CYCLES"]) / dr
    active = (t2["GRBM_GUI_ACTIVE"] - t1["GRBM_GUI_ACTIVE"]) / 8 / dr
    mops = (t2["SQ_INSTS_VALU_MFMA_MOPS_F32"] - t1["SQ_INSTS_VALU_MFMA_MOPS_F32"]) / dr
    kern = {}
    for k in k2:
        d = (n2[k] - n1.get(k, 0)) / dr
        if d <= 0:
            continue
        b = (k2[k]["SQ_VALU_MFMA_BUSY_CYCLES"] - k1.get(k, {}).get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)) / dr
        a = (k2[k]["GRBM_GUI_ACTIVE"] - k1.get(k, {}).get("GRBM_GUI_ACTIVE", 0.0)) / 8 / dr
        kern[k] = {"launches_per_rollout": d, "active_us_per_rollout_at_2.4GHz": a / 2400.0,
                   "mfma_util": b / (a * 1024) if a else None}
    res = {"command": "python3 tools/imag_bench.py cfg2 --replays {10,60} under rocprofv3 --pmc (difference of the two runs)",
           "T_img_ms_under_profiler": t_img_ms, "launches_per_rollout": sum(v["launches_per_rollout"] for v in kern.values()),
           "mfma_busy_cycles_per_rollout": busy, "dispatch_active_cycles_per_rollout": active,
           "mfma_busy_over_kernel_time": busy / (active * 1024),
           "mfma_busy_over_T_img_at_2.4GHz": busy / (t_img_ms * 1e-3 * 2.4e9 * 1024),
           "mfma_gflop_per_rollout": mops * 512 / 1e9, "kernels": kern}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "kernels"}, indent=1))


if __name__ == "__main__":
    main()
