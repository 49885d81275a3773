#!/usr/bin/env python3
"""Does a CU-masked HIP stream pay beside a chain of dependent few-row launches?  (MI355X only, development aid.)

    python tools/cumask_probe.py [--free 64]

1. a stream made by hipExtStreamCreateWithCUMask with `free` of the 256 CUs cleared: is the mask honoured eagerly, and by
   a hipGraph captured on / launched on that stream?  (big GEMM: time should rise by 256 / (256 - free));
2. a chain of dependent 16-row GEMMs (the observe scan's shape) on the main stream: alone, beside big GEMMs on an
   unmasked second stream, beside the same GEMMs on the masked stream.
"""
import argparse
import ctypes
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "dreamerv3-torch_amd"))
import torch  # noqa: E402

from dv3hip import ops  # noqa: E402


def hip():
    for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"):
        try:
            return ctypes.CDLL(name)
        except OSError:
            continue
    raise RuntimeError("libamdhip64 not found")


def masked_stream(free, order, invert=False):
    """Stream whose queue may use all CUs but `free` of them (invert: only those).  `order`: "low" clears the lowest mask bits, "high" the highest."""
    lib = hip()
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    words = (n_cu + 31) // 32
    bits = [1] * n_cu
    if order in ("low", "high"):
        rng = range(free) if order == "low" else range(n_cu - free, n_cu)
    elif order == "group8":
        # groups of 8 mask bits, every (256 / free)-th group cleared: symmetric over the XCDs whether the mask's bits run
        # XCD-major (bit -> XCD bit // 32) or XCD-interleaved (bit -> XCD bit % 8)
        every = n_cu // free
        rng = [i for i in range(n_cu) if (i // 8) % every == 0]
    elif order == "stride":
        every = n_cu // free
        rng = [i for i in range(n_cu) if i % every == 0]
    else:
        raise ValueError(order)
    for i in rng:
        bits[i] = 0
    if invert:
        bits = [1 - b for b in bits]
    mask = (ctypes.c_uint32 * words)()
    for i, b in enumerate(bits):
        if b:
            mask[i // 32] |= 1 << (i % 32)
    s = ctypes.c_void_p()
    rc = lib.hipExtStreamCreateWithCUMask(ctypes.byref(s), ctypes.c_uint32(words), mask)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask -> {rc}")
    return torch.cuda.ExternalStream(s.value)


def timed(fn, stream, reps=5):
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record(stream)
        fn()
        b.record(stream)
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--free", type=int, default=64)
    ap.add_argument("--order", default="low")
    args = ap.parse_args()
    dev = "cuda"
    # hipExtStreamCreateWithCUMask makes a BLOCKING stream: it synchronises implicitly with the NULL stream (torch's default
    # stream), so the main work has to live on a stream of its own to run beside it
    main_s = torch.cuda.Stream()
    torch.cuda.set_stream(main_s)
    ms = masked_stream(args.free, args.order)
    plain = torch.cuda.Stream()

    A = torch.randn(4096, 4096, device=dev)
    B = torch.randn(4096, 4096, device=dev)
    C = torch.empty(4096, 4096, device=dev)

    def big(n=4):
        for _ in range(n):
            ops.gemm(A, B, C, transB=True)

    torch.cuda.synchronize()
    t_main = timed(lambda: big(), main_s)
    with torch.cuda.stream(ms):
        t_mask = timed(lambda: big(), ms)
    print(f"4 x 4096^3 GEMM eager: main stream {t_main:.3f} ms, masked stream ({args.free} CUs free, {args.order}) {t_mask:.3f} ms "
          f"-> ratio {t_mask / t_main:.3f} (expected {256 / (256 - args.free):.3f} when honoured)")

    # graph captured on the masked stream, replayed on it
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(ms):
        big()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=ms):
            big()
        t_gmask = timed(lambda: g.replay(), ms)
    t_gmain = timed(lambda: g.replay(), main_s)
    print(f"same, hipGraph captured on the masked stream: replayed on it {t_gmask:.3f} ms, replayed on the main stream {t_gmain:.3f} ms")

    # the dependent chain
    x = torch.randn(16, 1024, device=dev)
    W = torch.randn(1536, 1024, device=dev) * 0.03
    W2 = torch.randn(1024, 1536, device=dev) * 0.03
    y = torch.empty(16, 1536, device=dev)

    def chain(n=200):
        for _ in range(n):
            ops.gemm(x, W, y, transB=True)
            ops.gemm(y, W2, x, transB=True)

    gc = torch.cuda.CUDAGraph()
    chain(4)
    torch.cuda.synchronize()
    with torch.cuda.graph(gc):
        chain()
    torch.cuda.synchronize()
    t_alone = timed(lambda: gc.replay(), main_s)
    print(f"chain of 400 dependent 16-row GEMMs (graph): alone {t_alone:.3f} ms = {t_alone * 1000 / 400:.2f} us/launch")

    gb = torch.cuda.CUDAGraph()
    with torch.cuda.stream(plain):
        with torch.cuda.graph(gb, stream=plain):
            big(3)
    for label, side in (("unmasked", plain), ("masked", ms)):
        graph_side = gb if side is plain else g

        def both():
            side.wait_stream(main_s)
            with torch.cuda.stream(side):
                graph_side.replay()
            gc.replay()
            main_s.wait_stream(side)

        t_both = timed(both, main_s)
        with torch.cuda.stream(side):
            t_side = timed(lambda: graph_side.replay(), side)

        # the chain's own time while the side work runs
        ca, cb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        side.wait_stream(main_s)
        with torch.cuda.stream(side):
            graph_side.replay()
            graph_side.replay()
        ca.record(main_s)
        gc.replay()
        cb.record(main_s)
        torch.cuda.synchronize()
        print(f"chain beside big GEMMs on the {label} stream: {t_both:.3f} ms (side work alone {t_side:.3f} ms, chain alone {t_alone:.3f} ms, "
              f"serial {t_side + t_alone:.3f} ms); chain's own time under contention {ca.elapsed_time(cb):.3f} ms")

    # the chain on the complementary mask: alone, and beside the masked side work
    cs = masked_stream(args.free, args.order, invert=True)
    with torch.cuda.stream(cs):
        t_c_alone = timed(lambda: gc.replay(), cs)

        def both_c():
            ms.wait_stream(cs)
            with torch.cuda.stream(ms):
                g.replay()
            gc.replay()
            cs.wait_stream(ms)

        t_c_both = timed(both_c, cs)
    print(f"chain on the complementary mask ({args.free} CUs): alone {t_c_alone:.3f} ms = {t_c_alone * 1000 / 400:.2f} us/launch; beside the masked side work "
          f"{t_c_both:.3f} ms (side alone {t_gmask:.3f})")

    # eager side work, graph chain
    for label, side in (("unmasked", plain), ("masked", ms)):
        def both():
            side.wait_stream(main_s)
            with torch.cuda.stream(side):
                big(3 if side is plain else 4)
            gc.replay()
            main_s.wait_stream(side)

        t_both = timed(both, main_s)
        print(f"chain (graph) beside EAGER big GEMMs on the {label} stream: {t_both:.3f} ms")


if __name__ == "__main__":
    main()
