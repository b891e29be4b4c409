cd /root/repo
L=/root/repo/2g-gcn_amd
echo "##### diag rebuilt with the wave index in a SCALAR register (readfirstlane): the predicated X3 P2 loop, 40 bytes of scratch"
export TWOG_LIB_PATH=$L/lib2ggcn_hip_diag.so
timeout 300 python3 tools/persist_stress.py 16 120 2 9 64 3 2>&1 | grep "^lib\|^run\|hs_o"| cut -c1-300
timeout 300 python3 tools/persist_stress.py 32 120 2 8 64 3 2>&1 | grep "^lib\|^run\|hs_o"| cut -c1-300
timeout 300 python3 tools/persist_stress.py 6 120 2 9 256 3 2>&1 | grep "^lib\|^run\|hs_o"| cut -c1-300
timeout 300 python3 tools/persist_stress.py 3 120 2 9 128 3 1 2>&1 | grep "^lib\|^run\|hs_o"| cut -c1-300
timeout 300 python3 tools/persist_stress.py 8 120 2 4 512 3 2>&1 | grep "^lib\|^run\|hs_o"| cut -c1-300
echo "##### seg_persist bench: diag (X3 P2) vs shipped"
timeout 300 python3 tools/seg_persist_bench.py 2>&1 | tail -12
unset TWOG_LIB_PATH
timeout 300 python3 tools/seg_persist_bench.py 2>&1 | tail -12
echo "##### shipped lib: persistent tests"
timeout 1500 python3 -m pytest tests/test_kernels_gpu.py -q -x -k "persistent or jitter" 2>&1 | tail -5
python3 bench.py --workload c2 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('c2', d['ms_per_step'], d.get('recurrence_paths'))"
python3 bench.py --workload c5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('c5', d['ms_per_step'], d.get('recurrence_paths'))"
