mkdir -p gpurun_out/final_stress
for t in callers charmodel epoch multitext pernet shapes shards text; do
  for seed in 9701 9702; do
    timeout 240 python tools/gpu_stress_$t.py $seed > gpurun_out/final_stress/${t}_$seed.log 2>&1
    echo "stress $t $seed rc $? : $(tail -1 gpurun_out/final_stress/${t}_$seed.log | cut -c1-160)"
  done
done
for seed in 9711 9712 9713 9714 9715 9716; do
  timeout 300 python tools/gpu_fuzz_api.py $seed 10 > gpurun_out/final_stress/fuzz_$seed.log 2>&1
  echo "fuzz $seed rc $? : $(tail -1 gpurun_out/final_stress/fuzz_$seed.log | cut -c1-160)"
done
for seed in 9751 9752; do
  FUZZ_LARGE=0.6 timeout 400 python tools/gpu_fuzz_api.py $seed 8 > gpurun_out/final_stress/fuzzL_$seed.log 2>&1
  echo "fuzz large $seed rc $? : $(tail -1 gpurun_out/final_stress/fuzzL_$seed.log | cut -c1-160)"
done
