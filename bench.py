#!/usr/bin/env python3
"""Benchmark of the MI355X-native DreamerV3 world-model training hot path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg2]

A "step" is ONE full training update of the reference's Dreamer._train (dreamer.py:192-200) on one
synthetic replay minibatch already resident in HBM: world-model forward/backward (CNN encoder, RSSM
observe scan over T, decoder + reward + cont heads, KL), its clip+Adam step, then the behaviour
update on the updated world model (imagination over H, lambda-returns, actor and critic
forward/backward, two clip+Adam steps) -- and, for N > 1, one gradient all-reduce (RCCL) per
optimizer.  The metric is BASELINE.json's: imagination-steps/s = N_gpus * B*T*H / time per update.

Prints ONE JSON line on rank 0 with the `roofline` (dominant MFMA kernel, measured live with HIP
events on the launch stream) and `cpu_baseline` (the CPU oracle timed on this box's host cores)
objects described in DESIGN.md.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(REPO, "dreamerv3-torch_amd")
for _p in (REPO, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 matrix = 256 CU x 256 flop/clk x 2.4 GHz
METRIC = "imagination-steps/sec"
WORKLOADS = {  # BASELINE.json configs
    "cfg1": "dmc_proprio walker_walk (MLP encoder/decoder) synthetic replay",
    "cfg2": "dmc_vision 64x64x3 synthetic replay",
    "cfg3": "atari100k 64x64 discrete-action synthetic replay",
    "cfg4": "dmc_vision XL (crafter-size model, cnn_depth 96) 64x64x3 synthetic replay",
    "cfg5": "crafter 64x64 synthetic replay, long sequences (per-GPU shard of the 128-sequence batch at DP=8)",
}


def synthetic_batch(name, seed, device):
    """SURVEY.md 8(d) synthetic replay minibatch (dv3hip.shapes.synthetic_batch), staged on the device before timing."""
    from dv3hip import shapes

    return {k: torch.from_numpy(v).to(device) for k, v in shapes.synthetic_batch(name, seed).items()}


def host_cores() -> int:
    """CPU share of this process: cgroup quota if set, else affinity, capped at 16 (the GPU box gives one
    GPU's job a 16-core share; asking torch for more threads than that oversubscribes and stalls)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 16))


def cpu_baseline(name, updates=2):
    """The oracle (CPU port of the reference's path, parity-pinned by tests/golden) timed on this host."""
    from oracle import dv3_oracle as O
    from tests import helpers as Hh
    from tests.golden import common

    cores = host_cores()
    torch.set_num_threads(cores)
    print(f"[bench] cpu_baseline: timing the oracle on {cores} host threads ...", file=sys.stderr, flush=True)
    s = common.SHAPES[name]
    t_all = []
    Hh.oracle_update(name)  # warm-up (allocator, thread pool)
    for i in range(updates):
        t0 = time.perf_counter()
        Hh.oracle_update(name)
        t_all.append(time.perf_counter() - t0)
        print(f"[bench] cpu_baseline update {i}: {t_all[-1]:.2f} s", file=sys.stderr, flush=True)
    t = float(np.median(t_all))
    units = s["B"] * s["T"] * s["H"]
    out = {"value": units / t, "unit": "imagination-steps/s", "cores": cores, "kind": "port",
           "sample": f"{updates} full updates of {name} (after 1 warm-up), torch CPU fp32, {cores} threads; "
                     f"median {t:.2f} s/update (includes building the noise/batch arrays, < 2 %)",
           "sec_per_update": t}
    try:  # reference-CPU figure DERIVED from the port: ratio measured where both run
        r = json.load(open(os.path.join(REPO, "profiles", "r02_cpu_ratio.json")))
        ratio = float(r["ratio_reference_over_oracle"])
        out["reference_cpu_derived"] = {
            "value": units / (t * ratio), "unit": "imagination-steps/s", "ratio_reference_over_port": ratio,
            "source": "profiles/r02_cpu_ratio.json (tools/cpu_ratio.py in the build container, cfg2, 8 threads); "
                      "derived = port time x ratio, not measured on this host"}
    except Exception:
        pass
    return out


def phase_timers(wm, beh, data, H, t_upd_ms, reps=10):
    """T_img / T_beh of SURVEY 8(d): device time of one hipGraph replay each (median of `reps`)."""
    from dv3hip import ops

    def replay_ms(fn, on=None):
        """on: capture and replay on that stream (a 128-CU lane of engine.Lanes) instead of a whole-chip torch stream."""
        fn()  # warm (workspace allocation happens outside capture)
        torch.cuda.synchronize()
        st = on if on is not None else torch.cuda.Stream()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(st):
            with torch.cuda.graph(g, stream=st, capture_error_mode="thread_local"):
                fn()
            g.replay()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(st):
                a.record()
                g.replay()
                b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        return float(np.median(ts))

    post, _, _ = wm._train(data)
    ops.PROFILE.by_shape = False
    ops.PROFILE.start()
    beh._imagine_fwd(post, H)
    prof = ops.PROFILE.stop()
    gflop = sum(v["flops"] for v in prof.values()) / 1e9  # dense-equivalent: the gather layers priced as the Linear they replace
    gflop_mfma = sum(v["flops"] for k, v in prof.items() if k.startswith(("gemm_kernel", "conv"))) / 1e9
    t_img = replay_ms(lambda: beh._imagine_fwd(post, H))

    def behaviour():
        beh.train_fwd_bwd(post)
        beh.train_opt(allreduce=False)

    t_beh = replay_ms(behaviour)
    # the rollout where the pipelined update runs it: alone on one 128-CU lane (half of the chip's matrix pipes)
    t_img_lane = None
    try:
        from dv3hip import engine

        ln = engine.Lanes.get(data["action"].device)
        if ln is not None:
            t_img_lane = replay_ms(lambda: beh._imagine_fwd(post, H), on=ln.streams["side"])
    except Exception as e:  # (a device without CU-masked streams: the figure is simply absent)
        print(f"[bench] no lane timing of the rollout: {e}", file=sys.stderr)
    return {"T_upd_ms": t_upd_ms, "T_beh_ms": t_beh, "T_img_ms": t_img, "T_img_lane_ms": t_img_lane,
            "method": "hipGraph replay of the phase, HIP events, median of 10",
            "imagine_fwd_gflop": gflop, "imagine_fwd_gflop_mfma": gflop_mfma, "imagine_fwd_tflops": gflop / t_img}


def launch_name(runner, mode):
    if not runner.use_graph:
        return "eager"
    if mode == "serial" or runner._pipe is None:
        return "hipGraph replay, one update after the other"
    return ("hipGraph replay, two-update pipeline: behaviour phase of update k beside the world-model phase of update "
            f"k+1 ({runner._pipe['mode']}), stream of back-to-back updates as in one agent call (dreamer.py:95-97)")


def run_updates(runner, data, n, mode):
    """n updates on `data`: "pipelined" = the stream of back-to-back updates of one agent call (dreamer.py:95-97), the
    behaviour phase of each issued beside the next one's world-model phase and the last one by flush(); "pairs" = two
    updates per call + flush (what the reference's train_ratio gives at the dmc configs); "serial" = one whole update
    after the other."""
    if mode == "serial":
        for _ in range(n):
            runner.step(data)
    elif mode == "pairs":
        for i in range(n):
            runner.step_pipelined(data)
            if i % 2 == 1 or i == n - 1:
                runner.flush()
    else:
        for _ in range(n):
            runner.step_pipelined(data)
        runner.flush()


def warm_up(runner, data, n, mode):
    """n untimed updates; for the pipelined modes as many more (at most 4) as it takes until every hipGraph the timed
    region replays has been captured (two eager calls, the serial halves, then the pipelined segments)."""
    run_updates(runner, data, n, mode)
    extra = 0
    while (mode != "serial" and runner.use_graph and runner._pipe is None and runner.pipeline_wanted() and extra < 4):
        run_updates(runner, data, 2, mode)
        extra += 2
    return extra


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start N copies of this script, one per GPU, with the
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* environment torch.distributed.run would give them (the seam is the
    reference's single optimizer step, tools.py:765-768: one all-reduce per optimizer).  Runs BEFORE anything in this
    process initialises the GPU; rank 0 prints the JSON line; the exit code is the first non-zero child code."""
    import socket
    import subprocess

    # (the device count is checked by the ranks themselves, in main(): the parent touches no GPU API at all)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        code = p.wait()
        if code != 0 and rc == 0:
            rc = code
            for q in procs:  # a dead rank would leave the others waiting in a collective
                if q.poll() is None:
                    q.terminate()
    return rc


def secondary_config(name, device, steps, warmup, serial=False):
    """Another BASELINE config measured in the SAME run on rank 0: time per update over `steps` hipGraph replays (the
    same timed region as the headline: pipelined where UpdateRunner takes the pipeline at this shape), the serial time,
    T_img, and the dominant MFMA kernel's fraction."""
    import models
    import tools
    from dv3hip import ops, shapes
    from dv3hip.graph import UpdateRunner

    shape = shapes.SHAPES[name]
    B, T, H = shape["B"], shape["T"], shape["H"]
    torch.manual_seed(0)
    cfg = shapes.make_config(name, str(device))
    wm = models.WorldModel(shapes.obs_space(name), None, 0, cfg).to(device)
    beh = models.ImagBehavior(cfg, wm).to(device)
    wm.requires_grad_(False), beh.requires_grad_(False)
    data = synthetic_batch(name, seed=0, device=device)
    runner = UpdateRunner(wm, beh)
    mode = "serial" if serial else "pipelined"
    warm_up(runner, data, warmup, mode)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_updates(runner, data, steps, mode)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    loss = float(runner.wm_metrics["model_loss"])
    ms_serial = ms
    if runner._pipe is not None:
        warm_up(runner, data, 2, "serial")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_updates(runner, data, steps, "serial")
        torch.cuda.synchronize()
        ms_serial = (time.perf_counter() - t0) / steps * 1e3
    ops.PROFILE.by_shape = False
    ops.PROFILE.start()
    runner.step(data, eager=True)
    prof = ops.PROFILE.stop()
    eng = {k: v for k, v in prof.items() if v["flops"] > 0 and k.startswith(("gemm_kernel", "conv"))
           and "skinny16" not in k and "narrowN" not in k}
    dom = max(eng, key=lambda k: eng[k]["ms"])
    ach = eng[dom]["flops"] / (eng[dom]["ms"] * 1e-3) / 1e12
    tm = phase_timers(wm, beh, data, H, ms, reps=5)
    tot_fl = sum(v["flops"] for v in prof.values())
    return {"workload": f"{name}: {WORKLOADS.get(name, name)}, batch {B} x seq {T}, horizon {H}", "ms_per_step": ms,
            "value": B * T * H / (ms * 1e-3), "unit": "imagination-steps/s", "steps": steps, "warmup": warmup,
            "launch": launch_name(runner, mode), "ms_per_step_serial": ms_serial,
            "model_loss": loss, "update_gflop": tot_fl / 1e9,
            "update_frac": tot_fl / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
            "roofline": {"kernel": dom, "kernel_symbol": ops.kernel_symbol(dom), "achieved": ach,
                         "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_F32_MFMA_TFLOPS,
                         "avg_launch_us": eng[dom]["ms"] * 1e3 / eng[dom]["launches"]},
            "T_img_ms": tm["T_img_ms"], "T_beh_ms": tm["T_beh_ms"],
            "T_img_frac": tm["imagine_fwd_tflops"] / PEAK_F32_MFMA_TFLOPS,
            "T_img_frac_mfma_executed": tm["imagine_fwd_gflop_mfma"] / tm["T_img_ms"] / PEAK_F32_MFMA_TFLOPS}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly (no hipGraph replay)")
    ap.add_argument("--serial", action="store_true",
                    help="time whole updates one after the other (no two-update pipeline): r01-r03's timed region")
    ap.add_argument("--by-shape", action="store_true", help="roofline leg: key GEMM launches by (M,N,K) too")
    ap.add_argument("--plain", action="store_true",
                    help="only the warm-up and timed updates (no roofline / phase-timer / staging / cpu legs): the "
                         "command tools/run_trace.sh puts under rocprofv3 so that every traced launch belongs to an update")
    ap.add_argument("--also", default="cfg1,cfg3,cfg4,cfg5",
                    help="further BASELINE configs measured in the same run on one GPU, comma-separated ('none' = skip)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))  # one child process per GPU; nothing has touched the GPU yet
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: a line for fewer ranks than asked is never printed")
    if world > torch.cuda.device_count() and os.environ.get("DV3_DIST_BACKEND", "nccl") != "gloo":
        raise SystemExit(f"--gpus {world} but {torch.cuda.device_count()} device(s) visible (RCCL needs one device per "
                         "rank; DV3_DIST_BACKEND=gloo rehearses the N>1 path on fewer)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the MI355X hot path has no CPU implementation")
    dev_index = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world == 1 and os.environ.get("DV3_BENCH_PG1"):
        # rehearsal of the N > 1 stream topology on one GPU: a one-rank RCCL group whose three all-reduces per update are
        # really issued (DV3_FORCE_ALLREDUCE=1, development library): 16.47 ms with the CU lanes, 16.84 without
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29533", rank=0, world_size=1, device_id=device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("DV3_DIST_BACKEND", "nccl")  # "gloo": rehearsal of the N>1 path on one GPU
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    from dv3hip import ops, shapes
    import tools

    name = args.config
    shape = shapes.SHAPES[name]
    B, T, H = shape["B"], shape["T"], shape["H"]
    torch.manual_seed(0)  # identical random-init replica on every rank (reference init scheme, tools.py:890-946)
    import models

    cfg = shapes.make_config(name, str(device))
    wm = models.WorldModel(shapes.obs_space(name), None, 0, cfg).to(device)
    beh = models.ImagBehavior(cfg, wm).to(device)
    wm.requires_grad_(False), beh.requires_grad_(False)
    tools.default_rng(device, seed=1234 + rank)  # per-rank sampling stream
    data = synthetic_batch(name, seed=rank, device=device)

    from dv3hip.graph import UpdateRunner

    runner = UpdateRunner(wm, beh, use_graph=not args.no_graph)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    mode = "serial" if (args.serial or args.no_graph) else "pipelined"
    extra_warm = warm_up(runner, data, args.warmup, mode)
    sync()
    if os.environ.get("DV3_BENCH_LATE_STREAM"):  # rehearsal: the caller moves to a torch stream of its own (update in line)
        torch.cuda.set_stream(torch.cuda.Stream(device))
        run_updates(runner, data, 3, mode)
        sync()
    t0 = time.perf_counter()
    run_updates(runner, data, args.steps, mode)
    sync()
    elapsed = torch.tensor([time.perf_counter() - t0], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed.item())
    loss = float(runner.wm_metrics["model_loss"])
    launch = launch_name(runner, mode)
    if not np.isfinite(loss):
        raise SystemExit("non-finite loss in the timed region")

    # ---- roofline leg: per-launch HIP-event timing of one more (eager) update on the launch stream
    roofline = None
    if args.plain:
        if rank == 0:
            print(json.dumps({"metric": METRIC, "value": world * B * T * H * args.steps / elapsed, "n_gpus": world,
                              "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
                              "config": {"workload": name, "launch": launch}, "plain": True}))
        if world > 1:
            dist.destroy_process_group()
        return
    if rank != 0:
        runner.step(data, eager=True)  # every rank takes part in the profiled update's all-reduces
        torch.cuda.synchronize()
    if rank == 0:
        ops.PROFILE.by_shape = args.by_shape
        ops.PROFILE.start()
        runner.step(data, eager=True)
        prof = ops.PROFILE.stop()
        mf = {k: v for k, v in prof.items() if v["flops"] > 0}
        # MFMA-bound candidates: the tile-engine kernels.  The M = batch scan GEMM ("skinny16"/"narrowN") is a
        # weight-streaming, latency-bound kernel and is reported on its own below.
        tile_engine = {k: v for k, v in mf.items()
                       if k.startswith(("gemm_kernel", "conv")) and "skinny16" not in k and "narrowN" not in k}
        if not tile_engine:  # toy shapes: everything ran on the few-row kernels
            tile_engine = mf
        dom = max(tile_engine, key=lambda k: tile_engine[k]["ms"])
        d = tile_engine[dom]
        ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
        tot_fl, tot_ms = sum(v["flops"] for v in mf.values()), sum(v["ms"] for v in mf.values())
        sym = ops.kernel_symbol(dom)
        # HBM/fabric bytes and MFMA-busy come from hardware counters, which only a rocprofv3 --pmc run can read: they
        # are looked up in the newest committed summaries of the SAME command (profiles/rNN_pmc_*.json, written by
        # tools/pmc_traffic.py / tools/pmc_mfma.py) and are NOT measured inside this run (flagged below).
        traffic, traffic_src, mfma_pmc = None, None, None
        import glob

        for path in sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_traffic.json")), reverse=True):
            try:
                pmc = json.load(open(path))["kernels"]
            except Exception:
                continue
            if sym in pmc:
                traffic = pmc[sym]["bytes_per_launch"]
                traffic_src = (f"profiles/{os.path.basename(path)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate "
                               "passes, gfx950 x2 read correction)")
                break
        for path in sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_mfma.json")), reverse=True):
            try:
                pm = json.load(open(path))["kernels"]
            except Exception:
                continue
            if sym in pm:
                mfma_pmc = {"mfma_util": pm[sym]["mfma_util"], "source": f"profiles/{os.path.basename(path)} "
                            "(SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs))"}
                break
        # the timed region runs the two-update pipeline: every kernel on a 128-CU lane, some on another tile than on the
        # whole chip (ops._LANE_TILES).  Per-kernel durations there cannot be taken with event pairs (two queues run side
        # by side): the dominant kernel of the committed rocprofv3 trace of `bench.py --plain` is quoted beside this leg's
        lanes_row = None
        for path in sorted((q for q in glob.glob(os.path.join(REPO, "profiles", "r*_kernel_stats.csv")) if "serial" not in q),
                           reverse=True):
            try:
                import csv

                top = next(csv.DictReader(open(path)))
                lanes_row = {"kernel_symbol": top["Name"], "avg_launch_us": float(top["AverageNs"]) / 1e3,
                             "share_of_kernel_time_pct": float(top["Percentage"]),
                             "source": f"profiles/{os.path.basename(path)} (rocprofv3 --kernel-trace --stats of bench.py --plain: "
                                       "the pipelined timed region, every kernel on a 128-CU lane)", "measured_in_run": False}
                break
            except Exception:
                continue
        scan = {k: v for k, v in mf.items() if "skinny16" in k}
        scan_ms = sum(v["ms"] for v in scan.values())
        scan_n = sum(v["launches"] for v in scan.values())
        scan_bytes = sum(v["bytes"] for v in scan.values())
        roofline = {"bound": "mfma", "kernel": dom, "kernel_symbol": sym, "achieved": ach,
                    "measured_on": "one eager update on the whole chip after the timed region, HIP event pair per launch; its "
                                   "kernel trace is profiles/rNN_serial_kernel_stats.csv (bench.py --plain --serial), whose top "
                                   "row is this kernel",
                    "pipelined_trace_top_kernel": lanes_row,
                    "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_F32_MFMA_TFLOPS,
                    "traffic": traffic, "traffic_source": traffic_src, "traffic_measured_in_run": False,
                    "mfma_busy_counter": mfma_pmc,
                    "launches_per_update": d["launches"], "avg_launch_us": d["ms"] * 1e3 / d["launches"],
                    "flops_per_launch": d["flops"] / d["launches"],
                    "algorithmic_bytes_per_launch": d["bytes"] / d["launches"],
                    "update": {"gflop": tot_fl / 1e9, "ms": elapsed / args.steps * 1e3,
                               "achieved": tot_fl / (elapsed / args.steps) / 1e12, "unit": "TFLOP/s",
                               "frac": tot_fl / (elapsed / args.steps) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                               "note": "algorithmic FLOPs of every launch of one update (gather layers priced as the "
                                       "Linear they replace) / the timed region's time per update / fp32 MFMA peak"},
                    "all_mfma_kernels": {"achieved": tot_fl / (tot_ms * 1e-3) / 1e12,
                                         "gflop_per_update": tot_fl / 1e9, "ms_per_update": tot_ms},
                    "scan_gemm": {"bound": "hbm", "kernel": "gemm_skinny_kernel (M = batch rows of the observe scan)",
                                  "launches_per_update": scan_n, "avg_launch_us": scan_ms * 1e3 / max(scan_n, 1),
                                  "achieved": scan_bytes / max(scan_ms * 1e-3, 1e-12) / 1e9, "unit": "GB/s",
                                  "peak": 8000.0, "note": "weight stream served by L2 / Infinity Cache; latency-bound"},
                    "by_kernel": {k: {"ms": round(v["ms"], 3), "n": v["launches"],
                                      "tflops": round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 2)}
                                  for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])[:(40 if args.by_shape else 16)]}}

    # ---- SURVEY 8(d) timers: T_img (= _imagine forward only) and T_beh (= ImagBehavior._train) beside T_upd,
    # each captured into its own hipGraph and replayed (single rank only: T_beh contains the actor/critic
    # all-reduces, which stay outside capture)
    timers = None
    if world == 1:
        def timed(m, n):
            warm_up(runner, data, 2, m)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run_updates(runner, data, n, m)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e3

        # T_upd_ms: one whole update after the other (r01-r03's timed region); the pipelined figures: see run_updates
        t_serial = elapsed / args.steps * 1e3 if mode == "serial" else timed("serial", args.steps)
        t_pipe = elapsed / args.steps * 1e3 if mode == "pipelined" else None
        t_pairs = timed("pairs", args.steps // 2 * 2) if (mode == "pipelined" and runner._pipe is not None) else None
        timers = phase_timers(wm, beh, data, H, t_serial)
        timers["T_upd_pipelined_ms"], timers["T_upd_pairs_ms"] = t_pipe, t_pairs
        timers["value_uses"] = "T_upd_pipelined_ms" if (mode == "pipelined" and runner._pipe is not None) else "T_upd_ms"
        if roofline is not None:
            g_mfma = timers.pop("imagine_fwd_gflop_mfma")
            roofline_gflop = timers.pop("imagine_fwd_gflop")
            roofline["imagine_fwd"] = {
                "gflop_dense_equivalent": roofline_gflop, "ms": timers["T_img_ms"],
                "achieved": timers["imagine_fwd_tflops"], "unit": "TFLOP/s", "peak": PEAK_F32_MFMA_TFLOPS,
                "frac": timers.pop("imagine_fwd_tflops") / PEAK_F32_MFMA_TFLOPS,
                "gflop_mfma_executed": g_mfma,
                "frac_mfma_executed": g_mfma / timers["T_img_ms"] / PEAK_F32_MFMA_TFLOPS,
                "on_lane": (None if not timers.get("T_img_lane_ms") else {
                    "ms": timers["T_img_lane_ms"], "cus": 128, "peak": PEAK_F32_MFMA_TFLOPS / 2,
                    "achieved": roofline_gflop / timers["T_img_lane_ms"], "unit": "TFLOP/s",
                    "frac_of_lane_peak": roofline_gflop / timers["T_img_lane_ms"] / (PEAK_F32_MFMA_TFLOPS / 2),
                    "frac_mfma_executed_of_lane_peak": g_mfma / timers["T_img_lane_ms"] / (PEAK_F32_MFMA_TFLOPS / 2),
                    "note": "the same rollout alone on one 128-CU lane, where the pipelined update runs it (beside the "
                            "world-model phase of the next update on the other lane)"}),
                "note": "imagination rollout (H actor evaluations + H-1 img_steps on B*T rows; the discarded H-th "
                        "successor of models.py:546 is not computed).  frac = SURVEY 8(d) algorithmic FLOPs (the "
                        "one-hot gather layers priced as the Linear they replace) / T_img / peak; frac_mfma_executed "
                        "= what the MFMA kernels actually multiply / T_img / peak"}

    # ---- PCIe-inclusive rate (never `value`): every step stages a fresh HOST batch through the pinned,
    # double-buffered stager (dv3hip/staging.py), overlapped with the previous update
    if world == 1 and timers is not None:
        from dv3hip.staging import BatchStager

        host = {k: v.cpu().numpy() for k, v in data.items()}
        stager = BatchStager(device, overlap=os.environ.get("DV3_STAGE_OVERLAP", "1") != "0")  # ("0": upload in front)
        step_fn = runner.step if mode == "serial" else runner.step_pipelined
        # (uploads on the stream the update is issued on: no second queue beside its dependent launches)
        with torch.cuda.stream(runner.launch_stream() or torch.cuda.current_stream()):
            for _ in range(3):
                step_fn(stager.stage(host))
            runner.flush()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step_fn(stager.stage(host))
            runner.flush()
            torch.cuda.synchronize()
        timers["T_upd_host_staged_ms"] = (time.perf_counter() - t0) / args.steps * 1e3

    others = None
    also = [c for c in args.also.split(",") if c and c not in ("none", name)]
    if rank == 0 and world == 1 and also:
        # free this config's graphs and workspaces first: cfg 4 / cfg 5 need tens of GB
        del runner, wm, beh
        import gc

        gc.collect()
        torch.cuda.empty_cache()
        others = {}
        for c in also:
            big = shapes.SHAPES[c]["deter"] >= 2048
            print(f"[bench] {c} ...", file=sys.stderr, flush=True)
            others[c] = secondary_config(c, device, steps=5 if big else max(6, args.steps // 2), warmup=3 if big else 5,
                                         serial=args.serial)
            gc.collect()
            torch.cuda.empty_cache()

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(name)

    if rank == 0:
        units = world * B * T * H
        out = {
            "metric": METRIC, "value": units * args.steps / elapsed, "unit": "imagination-steps/s",
            "n_gpus": world, "ranks": dist.get_world_size() if world > 1 else 1,
            "backend": (dist.get_backend() if world > 1 else None), "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{name}: {WORKLOADS.get(name, name)}, RSSM deter={shape['deter']} hidden="
                                   f"{shape['hidden']} stoch={shape['stoch']}x{shape['discrete']}, "
                                   f"{'one-hot' if shape['actor_dist'] == 'onehot' else 'continuous'} actor ({shape['A']}), "
                                   f"imag_gradient {shape['imag_gradient']}, batch {B} x seq {T} per GPU, horizon {H}; "
                                   "one step = full Dreamer._train update (world model + actor + critic fwd/bwd, "
                                   "gradient all-reduce, 3x clip+Adam)",
                       "global_batch": B * world, "seq_len": T, "horizon": H, "parallelism": f"dp{world}",
                       "launch": launch},
            "warmup_extra_for_capture": extra_warm, "model_loss": loss, "timers": timers,
            "roofline": roofline, "cpu_baseline": cpu, "configs": others,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
