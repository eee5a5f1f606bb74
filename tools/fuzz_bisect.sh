#!/bin/bash
# Development aid (GPU box): one fuzzer seed under different switch settings.  usage: tools/fuzz_bisect.sh <seed> <FUZZ_LARGE> "ENV=.. ENV=.." ...
seed=$1; large=$2; shift 2
mkdir -p gpurun_out/bisect
i=0
for envs in "" "$@"; do
  i=$((i + 1))
  env $envs FUZZ_LARGE=$large timeout 400 python tools/gpu_fuzz_api.py $seed 10 > gpurun_out/bisect/fuzz_${seed}_$i.log 2>&1
  echo "[$envs] $(tail -1 gpurun_out/bisect/fuzz_${seed}_$i.log | cut -c1-80) | $(grep -o "MISMATCH.*\]: \[.*" gpurun_out/bisect/fuzz_${seed}_$i.log | sed 's/.*\]: //' | cut -c1-120)"
done
