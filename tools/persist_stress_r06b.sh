cd /root/repo
L=/root/repo/2g-gcn_amd
export TWOG_LIB_PATH=$L/lib2ggcn_hip_diag.so TWOG_STRESS_DETAIL=1
echo "##### diag: partial object rows at h = 512 (18 of 32 rows), one chunk"
timeout 300 python3 tools/persist_stress.py 2 20 2 9 512 2
echo "##### diag: full object rows (32 of 32) at h = 64, eight chunks"
timeout 300 python3 tools/persist_stress.py 32 20 2 8 64 2
echo "##### diag: partial rows h = 64, one chunk, detail"
timeout 300 python3 tools/persist_stress.py 2 6 2 9 64 2
echo "##### diag: 3 clips x 9 objects = 27 rows, h = 128"
timeout 300 python3 tools/persist_stress.py 3 6 2 9 128 2 1
