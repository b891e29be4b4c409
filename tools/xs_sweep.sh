#!/bin/bash
# Chain-GEMM launches in isolation: unsplit, the library's model, forced slice counts (tools/gemm_chain_bench.py)
out=gpurun_out/xs_sweep.txt
: > $out
echo "== unsplit (twog_gemm_f32)" >> $out
timeout 300 python3 tools/gemm_chain_bench.py >> $out 2>&1
echo "== chain, library's model" >> $out
timeout 300 python3 tools/gemm_chain_bench.py chain >> $out 2>&1
for s in 2 3 4 6 8 12; do
  echo "== chain, TWOG_GEMM_XSPLIT=$s" >> $out
  TWOG_GEMM_XSPLIT=$s timeout 300 python3 tools/gemm_chain_bench.py chain >> $out 2>&1
done
cat $out
