run() {  # workload, label, env...
  local w=$1; local label=$2; shift; shift
  env "$@" python3 bench.py --workload $w --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python3 -c "
import json,sys
l=[x for x in sys.stdin if x.startswith('{')][-1]; d=json.loads(l)
print('%-6s %-34s %7.2f ms  %7.1f clips/s' % ('$w', '$label', d['ms_per_step'], d['value']))"
}
for w in c2 c5; do
  run $w "default" A=1
  run $w "all GEMMs 64x64 class (TILE=64)" TWOG_GEMM_TILE=64
  run $w "no split-K (SPLITK=1)" TWOG_GEMM_SPLITK=1
  run $w "BIG_MIN=400" TWOG_GEMM_BIG_MIN=400
  run $w "default" A=1
done
