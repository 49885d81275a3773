#!/usr/bin/env python3
"""rocprofv3 PMC counter CSVs -> per-kernel memory-side traffic per launch.

    python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>

Inputs are two SEPARATE passes of the same command (TCC has 4 counter slots; FETCH_SIZE needs 3, WRITE_SIZE 2):
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d OUT -- python bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d OUT -- python bench.py ...
Units / corrections (MI355X_MICROARCH.md §HBM): both counters are in KiB; on gfx950 FETCH_SIZE reports exactly
half of the bytes of wide (16 B/lane) coalesced reads, which is what every kernel here issues, so it is
doubled; WRITE_SIZE is exact.  Infinity-Cache hits are counted, so this is fabric-side traffic, an upper
bound on HBM traffic.
"""
import collections
import csv
import json
import sys


def agg(path, counter):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        d[r["Kernel_Name"]][0] += 1
        d[r["Kernel_Name"]][1] += float(r["Counter_Value"])
    return d


def main():
    fetch, write, out = sys.argv[1:4]
    f, w = agg(fetch, "FETCH_SIZE"), agg(write, "WRITE_SIZE")
    res = {}
    for name, (n, v) in f.items():
        nw, vw = w.get(name, [0, 0.0])
        rd = 2.0 * v / n * 1024.0
        wr = (vw / nw * 1024.0) if nw else 0.0
        res[name] = {"launches": n, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
                     "bytes_per_launch": rd + wr, "fetch_size_kib_raw": v / n, "write_size_kib_raw": vw / max(nw, 1)}
    json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes); read = 2 x FETCH_SIZE x 1024 "
                       "(gfx950 wide-load correction), write = WRITE_SIZE x 1024; per launch averages",
               "kernels": res}, open(out, "w"), indent=1, sort_keys=True)
    top = sorted(res.items(), key=lambda kv: -kv[1]["bytes_per_launch"] * kv[1]["launches"])[:12]
    for k, v in top:
        print(f"{k[:90]:90s} {v['bytes_per_launch'] / 1e6:10.2f} MB/launch x {v['launches']}")


if __name__ == "__main__":
    main()
