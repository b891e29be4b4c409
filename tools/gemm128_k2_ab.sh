#!/bin/bash
# Round 6: 16-wave (two k-groups) 128x128 tile for launches of at most one tile per CU (TWOG_X3_K2, gemm_f32.hip). One box, alternating.
run() {  # label, env...
  local label=$1; shift
  env "$@" python3 bench.py --no-cpu-baseline --steps 15 --warmup 4 2>/dev/null | python3 -c "
import json,sys
l=[x for x in sys.stdin if x.startswith('{')][-1]; d=json.loads(l)
c=d.get('roofline_chain',{}).get('loops',{})
print('%-28s %7.2f ms  %7.1f clips/s  fwd %7.1f clips/s  frac %.4f  us/step: seg fwd %.1f bwd %.1f' % ('$label', d['ms_per_step'], d['value'], d.get('forward_only_clips_per_s', 0) or 0, d['roofline']['frac'], c['segrnn_fwd']['us_per_time_step'], c['segrnn_bwd']['us_per_time_step']))"
}
run "8-wave tile (K2=0)" TWOG_X3_K2=0
run "16-wave tile (K2=1)" TWOG_X3_K2=1
run "8-wave tile (K2=0)" TWOG_X3_K2=0
run "16-wave tile (K2=1)" TWOG_X3_K2=1
