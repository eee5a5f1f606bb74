#!/bin/bash
# Development aid (GPU box): rocprofv3 per-kernel averages of tools/gpu_steps.py for experimental builds
for n in "$@"; do
  echo "abl=$n"
  RECUR_AMD_LIB=$GRAFT_REPO_ROOT/build/dev/abl_$n/librecur_amd.so bash tools/kstats.sh 30 | grep "${KGREP:-text_top}"
done
