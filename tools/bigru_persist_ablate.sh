# Run on the GPU box: the persistent BiGRU forward (csrc/gru_persist.hip) with parts removed at build time
# (-DTWOG_GP_ABLATE=bits; wrong results by design, only the times count), bench shape.
# NOTE: the shipped kernel carries no measurement branches; the -DTWOG_GP_ABLATE hooks this script builds live in the
# tree of commit c2dc749 (`git show c2dc749:2g-gcn_amd/csrc/gru_persist.hip`), whose numbers are in
# profiles/r04_bigru_persistent.txt. Check that file out beside the shipped one to repeat them.
#   usage: bash tools/bigru_persist_ablate.sh "0 1 2 4 8 16 3 7 31"
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT/2g-gcn_amd/csrc"
for a in ${1:-0 1 2 4 8 16}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DTWOG_GP_ABLATE=$a -c gru_persist.hip -o /tmp/gp_$a.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 /tmp/gp_$a.o $(ls *.o | grep -v gru_persist.o) -o /tmp/lib_gp_$a.so
  echo "== TWOG_GP_ABLATE=$a"
  TWOG_LIB_PATH=/tmp/lib_gp_$a.so TWOG_PROBE_ONLY=1 python3 "$ROOT/tools/bigru_persist_probe.py" 2>&1 | grep "PERSIST=1" | head -2
done
