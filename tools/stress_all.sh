mkdir -p gpurun_out/final_stress
for t in callers charmodel epoch multitext pernet shapes shards text; do
  for seed in 9501 9502; do
    timeout 240 python tools/gpu_stress_$t.py $seed > gpurun_out/final_stress/${t}_$seed.log 2>&1
    echo "stress $t $seed rc $? : $(tail -1 gpurun_out/final_stress/${t}_$seed.log | cut -c1-160)"
  done
done
for seed in 9601 9602 9603 9604 9605 9606; do
  timeout 300 python tools/gpu_fuzz_api.py $seed 10 > gpurun_out/final_stress/fuzz_$seed.log 2>&1
  echo "fuzz $seed rc $? : $(tail -1 gpurun_out/final_stress/fuzz_$seed.log | cut -c1-160)"
done
for seed in 9651 9652; do
  FUZZ_LARGE=0.6 timeout 400 python tools/gpu_fuzz_api.py $seed 8 > gpurun_out/final_stress/fuzzL_$seed.log 2>&1
  echo "fuzz large $seed rc $? : $(tail -1 gpurun_out/final_stress/fuzzL_$seed.log | cut -c1-160)"
done
