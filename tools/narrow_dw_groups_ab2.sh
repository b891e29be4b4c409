run() {  # workload, label, env...
  local w=$1; local label=$2; shift; shift
  env "$@" python3 bench.py --workload $w --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python3 -c "
import json,sys
l=[x for x in sys.stdin if x.startswith('{')][-1]; d=json.loads(l)
print('%-12s %-22s %7.2f ms  %7.1f clips/s' % ('$w', '$label', d['ms_per_step'], d['value']))"
}
for w in defaults defaults_seg defaults defaults_seg defaults; do
  run $w "one by one" TWOG_BATCH_DW_NARROW_ROWS=0
  run $w "grouped (default)" A=1
done
