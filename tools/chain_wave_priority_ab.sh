#!/bin/bash
# Round 6 experiments on the BiGRU backward chain beside the side stream's dW GEMMs (one box, alternating):
#  - issue priority of the chain's waves (s_setprio; variant libraries built with -DTWOG_CHAIN_PRIO=n: see the profile text)
#  - one k-tile per barrier interval in the chain tiles (TWOG_X3S_KU=1): 48 KB of LDS per workgroup instead of 96 KB, so that a
#    chain workgroup fits beside TWO resident 128x128 workgroups (2 x 48 KB of the CU's 160 KB) instead of having to wait for one to retire
run() {  # label, env...
  local label=$1; shift
  env "$@" python3 bench.py --no-cpu-baseline --steps 15 --warmup 4 2>/dev/null | python3 -c "
import json,sys
l=[x for x in sys.stdin if x.startswith('{')][-1]; d=json.loads(l)
c=d.get('roofline_chain',{}).get('loops',{})
print('%-36s %7.2f ms  %7.1f clips/s  frac %.4f  us/step: bigru fwd %.1f bwd %.1f  seg fwd %.1f bwd %.1f' % ('$label', d['ms_per_step'], d['value'], d['roofline']['frac'], c['bigru_fwd']['us_per_time_step'], c['bigru_bwd']['us_per_time_step'], c['segrnn_fwd']['us_per_time_step'], c['segrnn_bwd']['us_per_time_step']))"
}
R=$PWD/2g-gcn_amd
run "shipped" A=1
run "KU=1 chain tiles" TWOG_X3S_KU=1
[ -f $R/lib2ggcn_hip_prio3.so ] && run "KU=1 + s_setprio 3" TWOG_X3S_KU=1 TWOG_LIB_PATH=$R/lib2ggcn_hip_prio3.so
run "KU=1 + low-priority side stream" TWOG_X3S_KU=1 TWOG_SIDE_PRIORITY=low
run "shipped" A=1
run "KU=1 chain tiles" TWOG_X3S_KU=1
run "KU=1, no side stream" TWOG_X3S_KU=1 TWOG_SIDE_DW=0
run "shipped, no side stream" TWOG_SIDE_DW=0
