# GPU box: cache policy of the X3 chain kernels' operand loads -- builds variants beside the shipped library (A streamed
# with nt so that it does not evict the weights from the XCD's L2 between the launches of a chain) and times the bs64 step.
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for v in "a2:-DTWOG_X3S_A_AUX=2" "a2b0:-DTWOG_X3S_A_AUX=2 -DTWOG_X3S_B_AUX=0" "b2:-DTWOG_X3S_B_AUX=2"; do
  tag=${v%%:*}; flags=${v#*:}
  ( cd 2g-gcn_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $flags -c gemm_f32.hip -o /tmp/gemm_$tag.o \
    && /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 /tmp/gemm_$tag.o $(ls *.o | grep -v gemm_f32.o) -o ../../gpurun_out/lib_$tag.so )
done
for rep in 1 2; do
  echo "== shipped"; python3 bench.py --steps 10 --no-cpu-baseline 2>&1 >/dev/null | grep "timed region"
  for tag in a2 b2; do
    echo "== $tag"; TWOG_LIB_PATH=$PWD/gpurun_out/lib_$tag.so python3 bench.py --steps 10 --no-cpu-baseline 2>&1 >/dev/null | grep "timed region"
  done
done
