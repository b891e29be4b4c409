cd /root/repo
L=/root/repo/2g-gcn_amd
export TWOG_LIB_PATH=$L/lib2ggcn_hip_diagj.so
for m in 0 15 0 3; do
echo "##### diagj, the SAME binary, TWOG_JITTER_MASK=$m"
TWOG_JITTER_MASK=$m timeout 300 python3 tools/persist_stress.py 16 20 2 9 64 3 2>&1 | grep "^lib\|^run"
TWOG_JITTER_MASK=$m timeout 300 python3 tools/persist_stress.py 32 20 2 8 64 3 2>&1 | grep "^lib\|^run"
done
unset TWOG_LIB_PATH
echo "##### tests"
timeout 1500 python3 -m pytest tests/test_kernels_gpu.py -q -x -k "jitter" 2>&1 | tail -5
( time timeout 1500 python3 -m pytest tests/test_parity_gpu.py -q -x -k "c5_hs512" 2>&1 | tail -5 ) 2>&1
TWOG_FUZZ_VERBOSE=1 python3 tools/parity_fuzz.py 287 405 286 2>&1 | tail -2 | cut -c1-1500
