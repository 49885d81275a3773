#!/usr/bin/env python3
"""rocprofv3 PMC counter CSV (SQ pass) -> per-kernel MFMA utilisation from hardware counters.

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_ANY \
              SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d OUT -- python3 bench.py ...
    python tools/pmc_mfma.py OUT/**/*_counter_collection.csv out.json [--simds 512]

--simds: SIMDs the profiled command's kernels can run on (default 1024 = the whole chip).  `bench.py --plain` runs the
two-update pipeline in "lanes" mode: EVERY kernel is dispatched by a queue that owns 128 of the 256 compute units
(512 SIMDs), so its busy share is taken over 512.  `bench.py --plain --serial` runs one update after the other on the
whole chip, except the reverse observe scan (scan_* kernels) and the weight gradients deferred beside it (conv_wgrad*,
the decoder's / heads' gemm_direct_tn launches), which run on 128-CU lanes: for those the JSON carries both figures
(`mfma_util` over the --simds count, `mfma_util_128cu` over 512) and `lane_kernel: true`.  gemm_tn_grouped_kernel is a mix there:
of its two grids per update the deferred cluster runs on a lane, the scan's on the whole chip (not flagged).

Per kernel (averages per launch):
  mfma_util   = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024)   -- matrix-pipe busy cycles summed over the
                chip's 1024 SIMDs over the cycles the dispatch was active (GRBM_GUI_ACTIVE is reported summed over
                the 8 XCDs, MI355X_MICROARCH.md 'DVFS give-back'); the same quantity as rocprofv3's derived MfmaUtil
  mfma_flops  = SQ_INSTS_VALU_MFMA_MOPS_F32 * 512  (rocprofv3's MfmaFlopsF32 expression)
  wait shares = SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES (all in quad-cycles)
"""
import collections
import csv
import json
import sys

SIMDS = 1024
XCDS = 8


LANE_KERNELS = ("scan_", "conv_wgrad", "obs_carry", "obs_blend_bwd")  # serial run: what the two CU-masked lanes execute


def main():
    global SIMDS
    src, out = sys.argv[1], sys.argv[2]
    if "--simds" in sys.argv:
        SIMDS = int(sys.argv[sys.argv.index("--simds") + 1])
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(int))
    for r in csv.DictReader(open(src)):
        k, c = r["Kernel_Name"], r["Counter_Name"]
        acc[k][c] += float(r["Counter_Value"])
        cnt[k][c] += 1
    res = {}
    for k, d in acc.items():
        n = max(cnt[k].values())
        g = d.get("GRBM_GUI_ACTIVE", 0.0)
        busy = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        wc = d.get("SQ_WAVE_CYCLES", 0.0)
        e = {"launches": n, "gui_active_cycles_per_launch": g / n / XCDS,
             "mfma_busy_cycles_per_launch": busy / n,
             "mfma_util": (busy / (g / XCDS * SIMDS)) if g else None,
             "mfma_util_128cu": (busy / (g / XCDS * 512)) if g else None,
             "lane_kernel": SIMDS == 512 or any(t in k for t in LANE_KERNELS),
             "mfma_flops_per_launch": d.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0) * 512 / n}
        if wc:
            e.update(wait_any=d.get("SQ_WAIT_ANY", 0.0) / wc, wait_inst_any=d.get("SQ_WAIT_INST_ANY", 0.0) / wc,
                     active_inst_any=d.get("SQ_ACTIVE_INST_ANY", 0.0) / wc)
        res[k] = e
    json.dump({"note": f"rocprofv3 --pmc SQ pass; mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * {SIMDS} "
                       "SIMDs); mfma_util_128cu = the same over the 512 SIMDs of a 128-CU lane (the figure that applies "
                       "where lane_kernel is true); per-launch averages over every dispatch of the kernel in the "
                       "profiled command", "simds": SIMDS,
               "kernels": res}, open(out, "w"), indent=1, sort_keys=True)
    top = sorted(res.items(), key=lambda kv: -kv[1]["mfma_busy_cycles_per_launch"] * kv[1]["launches"])[:14]
    for k, v in top:
        print(f"{k[:84]:84s} n={v['launches']:5d} util={100 * (v['mfma_util'] or 0):5.1f}% "
              f"({100 * (v['mfma_util_128cu'] or 0):5.1f}% of 128 CUs{', lane' if v['lane_kernel'] else ''})  "
              f"{v['mfma_flops_per_launch'] / 1e9:7.2f} GFLOP/launch  {v['gui_active_cycles_per_launch']:9.0f} cyc")


if __name__ == "__main__":
    main()
