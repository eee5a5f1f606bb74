#!/usr/bin/env python3
"""HBM-side bytes per launch of the dominant kernels, as bench.py's roofline block reads them
(profiles/rNN_pmc_traffic.json): from the same two rocprofv3 counter passes as
tools/pmc_summary.py, traffic = 2 x FETCH_SIZE + WRITE_SIZE (KB -> bytes; the factor 2 is the
gfx950 correction of MI355X_MICROARCH.md's HBM section).

    python tools/pmc_traffic_json.py <fetch_dir> <write_dir> > profiles/rNN_pmc_traffic.json
"""
import json
import sys

from pmc_summary import per_kernel

CLASSES = {"bptt_chain_gemm": ("k_chain_persist", "k_chain_main"), "delta_gemm": ("k_delta_direct", "k_delta_dma")}


def durations(d):
    """average kernel duration (us) by kernel name from the pass's kernel trace"""
    import csv
    import glob
    acc = {}
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            a = acc.setdefault(r["Kernel_Name"], [0, 0.0])
            a[0] += 1
            a[1] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3
    return {k: v[1] / max(v[0], 1) for k, v in acc.items()}


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    dur = durations(sys.argv[1])
    out = {}
    for cls, names in CLASSES.items():
        for k in fetch:
            if any(n in k for n in names):
                n, s = fetch[k]
                wn, ws = write.get(k, (0, 0.0))
                f, w = s / max(n, 1), ws / max(wn, 1)
                out[cls] = {"kernel": k.split("(")[0], "launches_profiled": n, "fetch_kb": f, "write_kb": w,
                            "bytes_per_launch": (2 * f + w) * 1024.0,
                            # what the kernel took in the counter pass: bench.py refuses the figure when the
                            # kernel it times has since changed by more than a fifth (a stale file)
                            "avg_us_when_profiled": dur.get(k)}
                break
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
    main()
