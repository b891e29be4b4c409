#!/bin/bash
# kernel-level profile of the bs8 (configs[1]) step:  bash tools/prof_c2.sh <tag>
TAG=${1:-rXX}
export TMPDIR=/tmp
rm -rf gpurun_out/prof_c2_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c2_$TAG -- python3 bench.py --workload c2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_c2_$TAG.log 2>&1
python3 tools/step_breakdown.py gpurun_out/prof_c2_$TAG > gpurun_out/${TAG}_bench_c2_step_breakdown.txt 2>&1
cp $(ls gpurun_out/prof_c2_$TAG/*/*_kernel_stats.csv | head -1) gpurun_out/${TAG}_bench_c2_kernel_stats.csv
rm -rf gpurun_out/prof_c2_$TAG
head -40 gpurun_out/${TAG}_bench_c2_step_breakdown.txt
