for v in 4 2 1 0; do
STT_AMD_LSTM_PREFETCH=$v python benchmarks/am_micro.py 3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('G=$v standalone lstm_ms', d['lstm_ms'])"
for i in 1 2; do
STT_AMD_LSTM_PREFETCH=$v python bench.py --no-cpu-baseline --steps 8 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); s=d['stage_ms_per_step']
print('G=$v ms/step %.3f'%d['ms_per_step'], 'RTF %.0f'%d['value'], {k:round(v,3) for k,v in s.items()})
"
done
done
