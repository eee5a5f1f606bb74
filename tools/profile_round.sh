#!/bin/bash
# GPU box: the profiles of one round, all from the SAME command (bench.py's default workload):
#   1. rocprofv3 --kernel-trace --stats            -> <out>/kernel_stats.csv + the bench line under the profiler
#   2. --pmc FETCH_SIZE, 3. --pmc WRITE_SIZE (separate passes: TCC has 4 slots, FETCH costs 3, WRITE 2)
#   4. --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE, 5. --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
# then tools/pmc_summary.py, tools/pmc_mfma_lds.py and tools/pmc_traffic_json.py turn them into the text /
# json summaries that are committed under profiles/.   usage: tools/profile_round.sh gpurun_out/r02prof
out=${1:-gpurun_out/prof}
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd "$root" || exit 1
rm -rf "$out" && mkdir -p "$out"
CMD="python3 bench.py --no-cpu-baseline --steps 60 --warmup 10"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o s -- $CMD > "$out/bench_under_rocprof.json" 2> "$out/trace.err"
cp $(find "$out/trace" -name 's_kernel_stats.csv' | head -1) "$out/kernel_stats.csv"
python3 tools/launch_boundaries.py $(find "$out/trace" -name 's_kernel_trace.csv' | head -1) k_fwd_fused 40 > "$out/launch_boundaries.txt"
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "mfma SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  set -- $pass; name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$out/$name" -o p -- $CMD > "$out/bench_$name.json" 2> "$out/$name.err"
  echo "pass $name rc=$?"
done
python3 tools/pmc_summary.py "$out/fetch" "$out/write" > "$out/pmc_fetch_write_per_kernel.txt"
python3 tools/pmc_mfma_lds.py "$out/mfma" "$out/lds" > "$out/pmc_mfma_lds_per_kernel.txt"
python3 tools/pmc_traffic_json.py "$out/fetch" "$out/write" > "$out/pmc_traffic.json"
rm -rf "$out/trace" "$out/fetch" "$out/write" "$out/mfma" "$out/lds"
head -12 "$out/kernel_stats.csv"; cat "$out/launch_boundaries.txt"; cat "$out/pmc_fetch_write_per_kernel.txt"; cat "$out/pmc_mfma_lds_per_kernel.txt"; cat "$out/pmc_traffic.json"
