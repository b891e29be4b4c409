# GPU box: the 128x128 X3 main loop with parts removed (build-time TWOG_X3_ABLATE=1..3, libraries built beside the real one
# by hand: hipcc -DTWOG_X3_ABLATE=n ... -o 2g-gcn_amd/lib_ablate<n>.so). Results of the ablated builds are WRONG by design;
# only the times matter. 1: planes stored without the split arithmetic; 2: no plane stores after the first k-tile;
# 3: neither stores nor fragment reads (MFMAs and barriers only: the unused global loads are eliminated too);
# 4: no stores, but the global loads are waited for and consumed.
cp 2g-gcn_amd/lib2ggcn_hip.so /tmp/real.so
echo "== shipped"; python3 tools/gemm_x3_bench.py 2>&1 | grep -E "NN|NT|TT|TN" | head -8
for a in ${ABL:-1 2 3 4}; do
  cp 2g-gcn_amd/lib_ablate$a.so 2g-gcn_amd/lib2ggcn_hip.so
  echo "== ablate $a"; python3 tools/gemm_x3_bench.py 2>&1 | grep -E "NN|NT|TT|TN" | head -8
done
cp /tmp/real.so 2g-gcn_amd/lib2ggcn_hip.so
