python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do
python bench.py --no-cpu-baseline --steps 8 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); s=d['stage_ms_per_step']
print('ms/step %.3f'%d['ms_per_step'], 'RTF %.0f'%d['value'], {k:round(v,3) for k,v in s.items()})
"
done
python benchmarks/batch_throughput.py 2>/dev/null | tail -2 | cut -c1-600
