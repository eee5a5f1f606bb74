#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 counter passes (one --pmc FETCH_SIZE, one
--pmc WRITE_SIZE, each with --kernel-trace --output-format csv).

    python tools/pmc_summary.py <fetch_dir> <write_dir> > profiles/rNN_pmc_fetch_write_per_kernel.txt

Raw counter values are in KB per launch.  MI355X_MICROARCH.md (HBM section): on gfx950
FETCH_SIZE reports half of the bytes of a wide coalesced read, so HBM-side read bytes =
2 x FETCH_SIZE; WRITE_SIZE is exact for 16-byte-per-lane stores.  traffic = 2*FETCH + WRITE.
"""
import csv
import glob
import sys
from collections import defaultdict


def per_kernel(d, counter):
    acc = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            a = acc[r["Kernel_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return acc


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    print("# rocprofv3 --kernel-trace --pmc FETCH_SIZE  and, in a separate pass,  --pmc WRITE_SIZE")
    print("# raw counter averages per launch, in KB; traffic = 2*FETCH + WRITE (gfx950 correction, see")
    print("# MI355X_MICROARCH.md, HBM section)")
    print("%-62s %8s %12s %12s %14s" % ("kernel", "launches", "FETCH_KB", "WRITE_KB", "traffic_MB"))
    rows = []
    for k in fetch:
        n, s = fetch[k]
        wn, ws = write.get(k, (0, 0.0))
        f, w = s / max(n, 1), ws / max(wn, 1)
        rows.append((2 * f + w, k, n, f, w))
    for t, k, n, f, w in sorted(rows, reverse=True):
        print("%-62s %8d %12.1f %12.1f %14.2f" % (k[:62], n, f, w, t / 1024))


if __name__ == "__main__":
    main()
