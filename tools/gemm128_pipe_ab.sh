#!/bin/bash
# Round 6: fragment reads one k-tile ahead of the MFMAs in the bf16x3 128x128 class (gemm_x3_pipe_kernel, TWOG_X3_PIPE). One box, alternating.
run() {  # label, env...
  local label=$1; shift
  env "$@" python3 bench.py --no-cpu-baseline --steps 15 --warmup 4 2>/dev/null | python3 -c "
import json,sys
l=[x for x in sys.stdin if x.startswith('{')][-1]; d=json.loads(l)
c=d.get('roofline_chain',{}).get('loops',{})
print('%-40s %7.2f ms  %7.1f clips/s  fwd %7.1f clips/s  frac %.4f  us/step: seg fwd %.1f bwd %.1f' % ('$label', d['ms_per_step'], d['value'], d.get('forward_only_clips_per_s', 0) or 0, d['roofline']['frac'], c['segrnn_fwd']['us_per_time_step'], c['segrnn_bwd']['us_per_time_step']))"
}
run "unpipelined (PIPE=0)" TWOG_X3_PIPE=0
run "one-tile-per-CU launches (PIPE=1)" TWOG_X3_PIPE=1
run "every forward-form launch (PIPE=3)" TWOG_X3_PIPE=3
run "all forms (PIPE=7)" TWOG_X3_PIPE=7
run "unpipelined (PIPE=0)" TWOG_X3_PIPE=0
run "one-tile-per-CU launches (PIPE=1)" TWOG_X3_PIPE=1
run "every forward-form launch (PIPE=3)" TWOG_X3_PIPE=3
run "all forms (PIPE=7)" TWOG_X3_PIPE=7
