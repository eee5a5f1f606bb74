#!/bin/bash
# Development aid (GPU box): per-kernel average durations of `tools/gpu_steps.py N` under rocprofv3.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT/tools" || exit 1
rm -rf ../gpurun_out/ks
rocprofv3 --kernel-trace --stats --output-format csv -d ../gpurun_out/ks -o s -- python3 gpu_steps.py "${1:-40}" > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("../gpurun_out/ks/**/s_kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print("   %-60s %6d %9.2f us %6s%%" % (r["Name"][:58], int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
rm -rf ../gpurun_out/ks
