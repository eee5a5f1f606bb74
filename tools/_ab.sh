for rep in 1 2 3 4; do
for sp in 0 20000; do
RECUR_AMD_SYNC_SPIN_US=$sp python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('spin $sp', round(d['value']), round(d['ms_per_step']*1000,2))"
done; done
for sp in 0 20000; do
RECUR_AMD_SYNC_SPIN_US=$sp python3 bench.py --gpus 1 --steps 300 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('300 steps, spin $sp', round(d['value']), round(d['ms_per_step']*1000,2))"
done
