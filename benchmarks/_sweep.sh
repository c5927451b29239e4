for a in 0 1 2 3 4 8 16 31; do
STT_AMD_ABLATE=$a timeout 120 python bench.py --no-cpu-baseline --steps 4 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); s=d['stage_ms_per_step']
print('ablate $a: dec %.3f'%s['decoder_next_ms'], d['decoder_phase_cycles_per_stream_step'])
"
done
