for sch in "" "16,48,48,48,48,26,16" "16,64,64,64,26,16" "32,64,64,48,26,16" "16,48,48,48,40,24,16,10" "24,56,56,56,34,16,8"; do
  echo "== $sch"
  STT_AMD_CHUNKS="$sch" timeout 300 python bench.py --no-cpu-baseline --steps 30 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['stage_ms_per_step'])"
done
