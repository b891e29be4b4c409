# GPU box: where a k-tile of the X3 chain kernel spends its cycles. Builds a DIAGNOSTIC library beside the shipped one
# (-DTWOG_STAMPS: s_memtime stamps at the phase boundaries of gemm_mainloop_x3s, summed per wave of workgroup 0), runs one
# chain shape through it and prints cycles per k-tile and phase. The stamps perturb the schedule (each one drains the wave's
# LDS counter): the phase SHARES are what to read, not the total.
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( cd 2g-gcn_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DTWOG_STAMPS -c gemm_f32.hip -o /tmp/gemm_stamps.o \
  && /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 /tmp/gemm_stamps.o $(ls *.o | grep -v gemm_f32.o) -o ../../gpurun_out/lib_stamps.so )
TWOG_LIB_PATH=$PWD/gpurun_out/lib_stamps.so TWOG_GEMM_BPLANES=0 python3 tools/stamps_probe.py
