#!/bin/bash
# Development aid (GPU box): HIP runtime API call counts of a python tool: tools/api_counts.sh tools/x.py args...
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd "$root" || exit 1
rm -rf gpurun_out/apic
rocprofv3 --hip-runtime-trace --stats --output-format csv -d gpurun_out/apic -o s -- python3 "$@" > /dev/null 2>&1
f=$(find gpurun_out/apic -name 's_hip_api_stats.csv' | head -1)
head -25 "$f" | cut -c1-120
rm -rf gpurun_out/apic
