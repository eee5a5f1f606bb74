#!/bin/bash
# Development aid (GPU box): rocprofv3 per-kernel averages of an arbitrary python tool: tools/kstats_cmd.sh <top n> tools/x.py args...
n=$1; shift
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd "$root" || exit 1
rm -rf gpurun_out/ksc
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ksc -o s -- python3 "$@" > /dev/null 2>&1
python3 - "$n" <<'PY'
import csv, glob, sys
f = glob.glob("gpurun_out/ksc/**/s_kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:int(sys.argv[1])]:
    print("   %-64s %6d %9.2f us %6s%%" % (r["Name"][:62], int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
rm -rf gpurun_out/ksc
