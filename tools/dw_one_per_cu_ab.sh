#!/bin/bash
# Round 6 experiment: the chain kernels of the BiGRU backward pass use 234-250 registers per lane (8 waves = the whole register
# file of a CU with two waves per SIMD), the dW GEMMs beside them 2 workgroups x 4 waves/SIMD x 128 = the whole file too: a chain
# workgroup can only be placed on a CU that BOTH GEMM workgroups have left. TWOG_DW_ONE_PER_CU=1 runs every dW launch with one
# workgroup per CU (half the file free), TWOG_GEMM_KS=0 makes the chain tiles 4-wave (one wave per SIMD: fits into that half).
run() {  # label, env...
  local label=$1; shift
  env "$@" python3 bench.py --no-cpu-baseline --steps 15 --warmup 4 2>/dev/null | python3 -c "
import json,sys
l=[x for x in sys.stdin if x.startswith('{')][-1]; d=json.loads(l)
c=d.get('roofline_chain',{}).get('loops',{})
print('%-44s %7.2f ms  %7.1f clips/s  frac %.4f  us/step: bigru fwd %.1f bwd %.1f  seg fwd %.1f bwd %.1f' % ('$label', d['ms_per_step'], d['value'], d['roofline']['frac'], c['bigru_fwd']['us_per_time_step'], c['bigru_bwd']['us_per_time_step'], c['segrnn_fwd']['us_per_time_step'], c['segrnn_bwd']['us_per_time_step']))"
}
run "shipped" A=1
run "4-wave chain tiles (KS=0)" TWOG_GEMM_KS=0
run "dW one workgroup per CU" TWOG_DW_ONE_PER_CU=1
run "dW one per CU + 4-wave chain tiles" TWOG_DW_ONE_PER_CU=1 TWOG_GEMM_KS=0
run "shipped" A=1
run "dW one per CU + 4-wave chain tiles" TWOG_DW_ONE_PER_CU=1 TWOG_GEMM_KS=0
run "no side stream" TWOG_SIDE_DW=0
