#!/bin/bash
# Development aid (GPU box): per-kernel average durations of `tools/gpu_callers_rate.py classify|multi|rnnca` under rocprofv3;
# the summary stays in gpurun_out/callers_<which>_kernel_stats.csv.
which="${1:-rnnca}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT/tools" || exit 1
rm -rf ../gpurun_out/ksc
rocprofv3 --kernel-trace --stats --output-format csv -d ../gpurun_out/ksc -o s -- python3 gpu_callers_rate.py "$which" 2>&1 | grep configs
python3 - "$which" <<'PY'
import csv, glob, shutil, sys
f = glob.glob("../gpurun_out/ksc/**/s_kernel_stats.csv", recursive=True)[0]
shutil.copy(f, "../gpurun_out/callers_%s_kernel_stats.csv" % sys.argv[1])
for r in list(csv.DictReader(open(f)))[:20]:
    print("   %-60s %6d %9.2f us %6s%%" % (r["Name"][:58], int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
rm -rf ../gpurun_out/ksc
