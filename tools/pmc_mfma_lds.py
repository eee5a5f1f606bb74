#!/usr/bin/env python3
"""Per-kernel matrix-pipe and LDS figures from rocprofv3 counter passes (each pass:
--kernel-trace --pmc <counters> --output-format csv, same command):

    python tools/pmc_mfma_lds.py <dir of the MFMA pass> <dir of the LDS pass> > profiles/rNN_pmc_mfma_lds_per_kernel.txt

MFMA pass: SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE.  rocprofv3 reports GRBM_GUI_ACTIVE as the SUM over
the 8 XCDs (MI355X_MICROARCH.md, DVFS give-back), so cycles = GRBM_GUI_ACTIVE / 8; GHz = cycles / kernel
duration = the clock the kernel ran at; MfmaUtil = busy / (cycles x 4 SIMDs x 256 CUs) = the share of
SIMD-cycles with the matrix pipe busy.  LDS pass: SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE (conflict cycles per
active LDS cycle).
"""
import csv
import glob
import sys
from collections import defaultdict


def load(d):
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    dur = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            a = acc[r["Kernel_Name"]][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            a = dur[r["Kernel_Name"]]
            a[0] += 1
            a[1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return acc, dur


XCDS = 8  # GRBM_GUI_ACTIVE comes summed over the XCDs


def avg(a):
    return a[1] / max(a[0], 1)


def main():
    m, mdur = load(sys.argv[1])
    l, _ = load(sys.argv[2])
    print("# rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE, and in a separate pass")
    print("# --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE; averages per launch")
    print("%-56s %8s %9s %12s %9s %9s %12s" % ("kernel", "launches", "avg_us", "cycles", "GHz", "MfmaUtil%", "LDS_conflict"))
    rows = []
    for k, c in m.items():
        if "GRBM_GUI_ACTIVE" not in c:
            continue
        act = avg(c["GRBM_GUI_ACTIVE"]) / XCDS
        busy = avg(c.get("SQ_VALU_MFMA_BUSY_CYCLES", [1, 0.0]))
        us = avg(mdur[k]) / 1e3 if k in mdur else 0.0
        lc = l.get(k, {})
        idx = avg(lc.get("SQ_LDS_IDX_ACTIVE", [1, 0.0]))
        conf = avg(lc.get("SQ_LDS_BANK_CONFLICT", [1, 0.0]))
        rows.append((us * c["GRBM_GUI_ACTIVE"][0], k, c["GRBM_GUI_ACTIVE"][0], us, act,
                     act / (us * 1e3) if us else 0.0, 100.0 * busy / (act * 4 * 256) if act else 0.0,
                     conf / idx if idx else 0.0))
    for _, k, n, us, act, ghz, util, conf in sorted(rows, reverse=True):
        print("%-56s %8d %9.2f %12.0f %9.2f %9.1f %12.3f" % (k[:56], n, us, act, ghz, util, conf))


if __name__ == "__main__":
    main()
