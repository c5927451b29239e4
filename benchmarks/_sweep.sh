python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python benchmarks/am_micro.py 3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('beam1', d['decoder_next_ms'], d['decoder_phase_cycles_per_stream_step'])"
python bench.py --no-cpu-baseline --steps 8 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); s=d['stage_ms_per_step']
print('ms/step %.3f'%d['ms_per_step'], {k:round(v,3) for k,v in s.items()}, d['decoder_phase_cycles_per_stream_step'])
"
