#!/bin/bash
# GPU box: a longer soak of the API call-sequence fuzzer on fresh seeds.  usage: tools/fuzz_soak.sh <first seed> <count> [FUZZ_LARGE]
first=${1:-9300}; count=${2:-20}; large=${3:-0.2}
mkdir -p gpurun_out/soak
for ((s = first; s < first + count; s++)); do
  FUZZ_LARGE=$large timeout 400 python tools/gpu_fuzz_api.py $s 10 > gpurun_out/soak/fuzz_$s.log 2>&1
  echo "fuzz $s (large $large) rc $? : $(tail -1 gpurun_out/soak/fuzz_$s.log | cut -c1-120) $(grep -c MISMATCH gpurun_out/soak/fuzz_$s.log) mismatches: $(grep -B40 MISMATCH gpurun_out/soak/fuzz_$s.log | grep -o '^trial [0-9]*: [a-z]*' | tail -1)"
done
