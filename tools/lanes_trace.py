#!/usr/bin/env python3
"""From a rocprofv3 kernel trace of `bench.py --plain`: which hardware queue ran what, and when, in the LAST update.

    python tools/lanes_trace.py <..._kernel_trace.csv> [out.txt]

Prints, per queue, the busy span and the kernel count of the last update, and the overlap of the two CU-masked lane queues
(the reverse observe scan beside the deferred weight gradients).  The update's boundaries are found from the Adam kernel
(3 launches per update: world model, actor, critic)."""
import collections
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    rows.sort(key=lambda r: r["s"])
    adam = [r for r in rows if "adam_kernel" in r["Kernel_Name"]]
    # last update = between the end of the third-last-but... Adam: updates end with the critic's Adam (every third launch)
    end = adam[-1]["e"]
    begin = adam[-4]["e"] if len(adam) >= 4 else rows[0]["s"]
    upd = [r for r in rows if begin < r["s"] <= end]
    t0 = upd[0]["s"]
    print(f"last update: {len(upd)} kernels, {(end - t0) / 1e6:.3f} ms from its first kernel's start to the last Adam's end", file=out)
    byq = collections.defaultdict(list)
    for r in upd:
        byq[r["Queue_Id"]].append(r)
    spans = {}
    for qid, ks in sorted(byq.items(), key=lambda kv: kv[1][0]["s"]):
        busy = sum(k["e"] - k["s"] for k in ks)
        a, b = ks[0]["s"], max(k["e"] for k in ks)
        spans[qid] = (a, b, ks)
        top = collections.Counter(k["Kernel_Name"].split("(")[0][:60] for k in ks).most_common(3)
        print(f"queue {qid}: {len(ks):4d} kernels, span {(a - t0) / 1e6:7.3f} .. {(b - t0) / 1e6:7.3f} ms, busy {busy / 1e6:6.3f} ms; "
              + "; ".join(f"{n} x{c}" for n, c in top), file=out)
    lanes = [q for q, (a, b, ks) in spans.items() if len(ks) < 0.5 * len(upd)]
    if len(lanes) >= 2:
        lanes.sort(key=lambda q: -len(spans[q][2]))
        (a0, b0, k0), (a1, b1, k1) = spans[lanes[0]], spans[lanes[1]]
        ov = max(0, min(b0, b1) - max(a0, a1))
        print(f"lane queues {lanes[0]} (the scan, {len(k0)} kernels) and {lanes[1]} (the deferred weight gradients, {len(k1)} kernels) "
              f"overlap for {ov / 1e6:.3f} ms of their {(b0 - a0) / 1e6:.3f} / {(b1 - a1) / 1e6:.3f} ms spans", file=out)
        main_q = max(spans, key=lambda q: len(spans[q][2]))
        inside = [k for k in spans[main_q][2] if k["s"] < max(b0, b1) and k["e"] > min(a0, a1)]
        print(f"kernels of the main queue inside the lanes' window: {len(inside)}", file=out)


def pipelined():
    """--pipelined: the two-update pipeline in the lanes schedule (bench.py --plain): the two lane queues carry the
    whole update.  Over the last four updates: each queue's busy time, and for how much of the window both / exactly
    one / none of them had a kernel running."""
    rows = list(csv.DictReader(open(sys.argv[1])))
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else sys.stdout
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    rows.sort(key=lambda r: r["s"])
    adam = [r for r in rows if "adam_kernel" in r["Kernel_Name"]]
    n_upd = 4
    end, begin = adam[-1]["e"], adam[-1 - 3 * n_upd]["e"]
    win = [r for r in rows if begin < r["s"] and r["e"] <= end]
    byq = collections.defaultdict(list)
    for r in win:
        byq[r["Queue_Id"]].append(r)
    qs = sorted(byq, key=lambda q: -len(byq[q]))[:2]
    print(f"last {n_upd} updates: {len(win)} kernels in {(end - begin) / 1e6:.3f} ms = {(end - begin) / 1e6 / n_upd:.3f} ms per update "
          f"(under the tracer); kernels per queue: " + ", ".join(f"queue {q}: {len(byq[q])}" for q in sorted(byq)), file=out)
    ev = []
    for i, q in enumerate(qs):
        busy = sum(k["e"] - k["s"] for k in byq[q])
        top = collections.Counter(k["Kernel_Name"].split("(")[0][:50] for k in byq[q]).most_common(3)
        print(f"queue {q}: busy {busy / 1e6:7.3f} ms ({100.0 * busy / (end - begin):4.1f} % of the window); "
              + "; ".join(f"{n} x{c}" for n, c in top), file=out)
        for k in byq[q]:
            ev.append((k["s"], 1 << i)), ev.append((k["e"], -(1 << i)))
    ev.sort()
    t, state, acc = begin, [0, 0], collections.Counter()
    for ts, d in ev:
        acc[(state[0] > 0) + (state[1] > 0)] += ts - t
        t = ts
        state[0 if abs(d) == 1 else 1] += 1 if d > 0 else -1
    acc[(state[0] > 0) + (state[1] > 0)] += end - t
    tot = float(end - begin)
    print(f"both lane queues busy {100 * acc[2] / tot:4.1f} % of the window, exactly one {100 * acc[1] / tot:4.1f} %, "
          f"none {100 * acc[0] / tot:4.1f} %", file=out)


if __name__ == "__main__":
    pipelined() if "--pipelined" in sys.argv else main()
