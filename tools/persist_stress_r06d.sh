cd /root/repo
L=/root/repo/2g-gcn_amd
echo "##### diag2: the X3 P2 loop with EVERY k-block slot multiplied (156 bytes of scratch per lane)"
export TWOG_LIB_PATH=$L/lib2ggcn_hip_diag2.so
timeout 300 python3 tools/persist_stress.py 16 20 2 9 64 3 2>&1 | grep "^lib\|^run"
timeout 300 python3 tools/persist_stress.py 32 20 2 8 64 3 2>&1 | grep "^lib\|^run"
timeout 300 python3 tools/persist_stress.py 3 6 2 9 128 3 1 2>&1 | grep "^lib\|^run"
timeout 300 python3 tools/persist_stress.py 16 120 2 9 64 3 2>&1 | grep "^lib\|^run"
echo "##### diag at h = 256 (4 of 8 k-block slots per wave in use)"
export TWOG_LIB_PATH=$L/lib2ggcn_hip_diag.so
timeout 300 python3 tools/persist_stress.py 6 20 2 9 256 3 2>&1 | grep "^lib\|^run"
unset TWOG_LIB_PATH
echo "##### c5_hs512"
timeout 1500 python3 -m pytest tests/test_parity_gpu.py -q -x -k "c5_hs512" 2>&1 | tail -40 | cut -c1-3000
timeout 900 python3 -m pytest tests/test_distributed_gpu.py -q -x 2>&1 | tail -15 | cut -c1-2000
