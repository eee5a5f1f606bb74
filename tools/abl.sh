#!/bin/bash
# Development aid (GPU box): bench classes for experimental builds build/dev/abl_<name>/librecur_amd.so
for n in "$@"; do
  echo -n "abl=$n "
  RECUR_AMD_LIB=$GRAFT_REPO_ROOT/build/dev/abl_$n/librecur_amd.so python bench.py --no-cpu-baseline --steps 60 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['roofline']['classes_ms_per_step']
print('%.0f/s  chain %.1f  delta %.1f  fwd %.1f  apply %.1f' % (d['value'], 1e3*c['bptt_chain_gemm'], 1e3*c['delta_gemm'], 1e3*c['forward_gemm'], 1e3*c['optimiser']))"
done
