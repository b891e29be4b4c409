# GPU box: where a step of the persistent segment-level forward launch spends its time. Builds a DIAGNOSTIC library beside the
# shipped one (-DTWOG_SP_STAMPS: wall-clock stamps at the phase boundaries of one workgroup per role), runs the BASELINE
# small-batch shapes through it and prints microseconds per step and phase.
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( cd 2g-gcn_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DTWOG_SP_STAMPS -c seg_persist.hip -o /tmp/segp_stamps.o \
  && /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 /tmp/segp_stamps.o $(ls *.o | grep -v seg_persist.o) -o ../../gpurun_out/lib_segp_stamps.so )
TWOG_LIB_PATH=$PWD/gpurun_out/lib_segp_stamps.so python3 tools/seg_persist_stamps.py
