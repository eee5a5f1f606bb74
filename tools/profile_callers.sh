#!/bin/bash
# GPU box: rocprofv3 --kernel-trace --stats of the callers' own workloads (tools/gpu_callers_rate.py: multi-head step,
# gstclassify generation, rnnca generation + frame fill) and of the reference's per-net call sequence at hidden 99
# (build/pernet_rate) and of rnn_char_epoch on the same net (tools/gpu_epoch_rate.py).  Host-side rates under the profiler are lower than unprofiled ones; the per-kernel averages are
# what these files are for.   usage: tools/profile_callers.sh gpurun_out/r02callers
out=${1:-gpurun_out/callers}
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd "$root" || exit 1
rm -rf "$out" && mkdir -p "$out"
for w in multi classify rnnca; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/t_$w" -o s -- python3 tools/gpu_callers_rate.py $w 60 > "$out/$w.log" 2>&1
  cp $(find "$out/t_$w" -name 's_kernel_stats.csv' | head -1) "$out/callers_${w}_kernel_stats.csv"
  rm -rf "$out/t_$w"
done
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/t_pernet" -o s -- ./build/pernet_rate -H 99 -d 30 -s 1000 > "$out/pernet.log" 2>&1
cp $(find "$out/t_pernet" -name 's_kernel_stats.csv' | head -1) "$out/callers_pernet_h99_kernel_stats.csv"
rm -rf "$out/t_pernet"
# text-predict's default configuration through the library's rnn_char_epoch (single-net branch on the device, then multi-tap)
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/t_epoch" -o s -- python3 tools/gpu_epoch_rate.py 99 30 3000 > "$out/epoch.log" 2>&1
cp $(find "$out/t_epoch" -name 's_kernel_stats.csv' | head -1) "$out/callers_epoch_h99_kernel_stats.csv"
rm -rf "$out/t_epoch"
head -8 "$out"/callers_*_kernel_stats.csv | cut -c1-150
