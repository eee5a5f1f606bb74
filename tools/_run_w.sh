show() { python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d['value']), round(d['ms_per_step']*1000,1), {k: round(v*1000,1) for k,v in d['roofline']['classes_ms_per_step'].items()})" "$1"; }
python -m pytest tests -m gpu -x -q -n 4 2>&1 | tail -3
for rep in 1 2 3; do
python bench.py --steps 100 2>/dev/null | show control_prefetch
done
