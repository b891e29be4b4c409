cd /root/repo
L=/root/repo/2g-gcn_amd
echo "##### case 286 under kernel variants"
for v in "" "TWOG_SEG_PERSIST=0" "TWOG_GEMM_X3=0" "TWOG_SEG_PERSIST=0 TWOG_GEMM_X3=0" "TWOG_BIGRU_PERSIST=0 TWOG_SEG_PERSIST=0"; do
  echo "== variant: $v"; env $v TWOG_FUZZ_VERBOSE=1 python3 tools/parity_fuzz.py 287 405 286 2>&1 | grep "off:\|FAIL" | cut -c1-900
done
echo "##### diag (spilling P2 variant, no jitter)"
export TWOG_LIB_PATH=$L/lib2ggcn_hip_diag.so
timeout 300 python3 tools/persist_stress.py 16 120 2 9 64 4
timeout 300 python3 tools/persist_stress.py 16 8 2 9 64 3
for mc in 1 2 3 4 6; do timeout 300 python3 tools/persist_stress.py $((2*mc)) 120 2 9 64 3 $mc; done
timeout 300 python3 tools/persist_stress.py 8 120 2 4 512 3
echo "##### diagj (spilling P2 variant + jitter)"
export TWOG_LIB_PATH=$L/lib2ggcn_hip_diagj.so
timeout 300 python3 tools/persist_stress.py 16 120 2 9 64 4
timeout 300 python3 tools/persist_stress.py 4 120 2 9 64 3 2
echo "##### jitter (shipped kernels + jitter)"
export TWOG_LIB_PATH=$L/lib2ggcn_hip_jitter.so
timeout 300 python3 tools/persist_stress.py 16 120 2 9 64 4
timeout 300 python3 tools/persist_stress.py 8 120 2 4 512 4
timeout 300 python3 tools/persist_stress.py 1 20 1 5 512 4
echo "##### shipped"
unset TWOG_LIB_PATH
timeout 300 python3 tools/persist_stress.py 16 120 2 9 64 4
