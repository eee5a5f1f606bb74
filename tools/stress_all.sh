#!/bin/bash
# GPU box: every randomised parity probe on fresh seeds (random shapes against the oracle, the API call-sequence fuzzer
# against the compiled reference).  usage: tools/stress_all.sh [base seed, default 9700]
b=${1:-9700}
mkdir -p gpurun_out/final_stress
for t in callers charmodel epoch multitext pernet shapes shards text; do
  for seed in $((b + 1)) $((b + 2)); do
    timeout 240 python tools/gpu_stress_$t.py $seed > gpurun_out/final_stress/${t}_$seed.log 2>&1
    echo "stress $t $seed rc $? : $(tail -1 gpurun_out/final_stress/${t}_$seed.log | cut -c1-160)"
  done
done
for seed in $((b + 11)) $((b + 12)) $((b + 13)) $((b + 14)) $((b + 15)) $((b + 16)); do
  timeout 300 python tools/gpu_fuzz_api.py $seed 10 > gpurun_out/final_stress/fuzz_$seed.log 2>&1
  echo "fuzz $seed rc $? : $(tail -1 gpurun_out/final_stress/fuzz_$seed.log | cut -c1-160)"
done
for seed in $((b + 51)) $((b + 52)); do
  FUZZ_LARGE=0.6 timeout 400 python tools/gpu_fuzz_api.py $seed 8 > gpurun_out/final_stress/fuzzL_$seed.log 2>&1
  echo "fuzz large $seed rc $? : $(tail -1 gpurun_out/final_stress/fuzzL_$seed.log | cut -c1-160)"
done
