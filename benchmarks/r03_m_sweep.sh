for cfg in "" "active=2" "pchunk=32" "pchunk=64" "pipeline=3"; do
  STT_AMD_TUNING="$cfg" timeout 200 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('$cfg', round(r['value']), round(r['ms_per_step'],3), r['verified'], {k:round(v,2) for k,v in r['stage_ms_per_step'].items()})"
done
