# Builds the measurement variants of the X3 main loop (see tools/x3_ablate.sh) into gpurun_out/, beside the shipped library.
set -e
cd "$(dirname "$0")/../2g-gcn_amd/csrc"
mkdir -p ../../gpurun_out
for a in ${ABL:-1 2 3 4}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DTWOG_X3_ABLATE=$a -c gemm_f32.hip -o /tmp/gemm_ablate$a.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 /tmp/gemm_ablate$a.o $(ls *.o | grep -v gemm_f32.o) -o ../../gpurun_out/lib_ablate$a.so
done
