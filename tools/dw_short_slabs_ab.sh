#!/bin/bash
# Round 6 experiment: SHORT dW workgroups (TWOG_GEMM_SLAB_K) with and without a low-priority side stream, so that the BiGRU
# backward chain finds free compute units sooner beside the side stream's weight-gradient GEMMs. One box, alternating.
run() {  # label, env...
  local label=$1; shift
  env "$@" python3 bench.py --no-cpu-baseline --steps 15 --warmup 4 2>/dev/null | python3 -c "
import json,sys
l=[x for x in sys.stdin if x.startswith('{')][-1]; d=json.loads(l)
c=d.get('roofline_chain',{})
print('%-44s %7.2f ms  %7.1f clips/s  frac %.4f  bigru_bwd us/step %s' % ('$label', d['ms_per_step'], d['value'], d['roofline']['frac'], c.get('loops',{}).get('bigru_bwd',{}).get('us_per_time_step')))"
}
run "shipped" A=1
run "low-priority side stream" TWOG_SIDE_PRIORITY=low
run "slabs of ~960 rows" TWOG_GEMM_SLAB_K=960
run "slabs of ~960 rows + low priority" TWOG_GEMM_SLAB_K=960 TWOG_SIDE_PRIORITY=low
run "slabs of ~1920 rows + low priority" TWOG_GEMM_SLAB_K=1920 TWOG_SIDE_PRIORITY=low
run "slabs of ~480 rows + low priority" TWOG_GEMM_SLAB_K=480 TWOG_SIDE_PRIORITY=low
run "shipped (again)" A=1
