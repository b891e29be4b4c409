run() {  # workload, label, env...
  local w=$1; local label=$2; shift; shift
  env "$@" python3 bench.py --workload $w --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python3 -c "
import json,sys
l=[x for x in sys.stdin if x.startswith('{')][-1]; d=json.loads(l)
print('%-6s %-26s %7.2f ms  %7.1f clips/s' % ('$w', '$label', d['ms_per_step'], d['value']))"
}
for w in c2 c5 c2 c5; do
  run $w "slabs + reduce (default)" A=1
  run $w "in-launch combine (LA=1)" TWOG_GEMM_LA=1
done
