# NOTE (round 4): the TWOG_X3_ABLATE branches were removed from csrc/gemm_f32.hip; this script applies to the round-3 tree
# (git 99b4ed3), whose results are committed as profiles/r03_x3_ablation.txt.
# GPU box: the 128x128 X3 main loop with parts removed (build-time TWOG_X3_ABLATE=1..4; each measurement library is built
# BESIDE the shipped one -- hipcc -DTWOG_X3_ABLATE=n ... -o gpurun_out/lib_ablate<n>.so, see tools/x3_ablate_build.sh -- and
# selected through TWOG_LIB_PATH: the shipped lib2ggcn_hip.so is never touched). Results of the ablated builds are WRONG by
# design; only the times matter. 1: planes stored without the split arithmetic; 2: no plane stores after the first k-tile;
# 3: neither stores nor fragment reads (MFMAs and barriers only: the unused global loads are eliminated too);
# 4: no stores, but the global loads are waited for and consumed.
echo "== shipped"; python3 tools/gemm_x3_bench.py 2>&1 | grep -E "NN|NT|TT|TN" | head -8
for a in ${ABL:-1 2 3 4}; do
  lib=gpurun_out/lib_ablate$a.so
  [ -f "$lib" ] || { echo "missing $lib (tools/x3_ablate_build.sh)"; continue; }
  echo "== ablate $a"; TWOG_LIB_PATH=$PWD/$lib python3 tools/gemm_x3_bench.py 2>&1 | grep -E "NN|NT|TT|TN" | head -8
done
