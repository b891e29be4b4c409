# GPU box: the X3 kernels with 6 chunk products (shipped) against a build with 8 (TWOG_X3_PRODUCTS=8: adds m l and l m) and
# against the native fp32-MFMA kernels (TWOG_GEMM_X3=0) on same-sign / post-ReLU / wide-exponent operands
# (tools/x3_bias_probe.py). The 8-product library is built beside the shipped one and selected through TWOG_LIB_PATH.
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( cd 2g-gcn_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DTWOG_X3_PRODUCTS=8 -c gemm_f32.hip -o /tmp/gemm_p8.o \
  && /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 /tmp/gemm_p8.o $(ls *.o | grep -v gemm_f32.o) -o ../../gpurun_out/lib_x3p8.so )
echo "== 6 products (shipped)"; python3 tools/x3_bias_probe.py
echo "== 8 products (TWOG_X3_PRODUCTS=8 build)"; TWOG_LIB_PATH=$PWD/gpurun_out/lib_x3p8.so python3 tools/x3_bias_probe.py
echo "== native fp32 MFMA (TWOG_GEMM_X3=0)"; TWOG_GEMM_X3=0 TWOG_GEMM_XSPLIT=1 python3 tools/x3_bias_probe.py
