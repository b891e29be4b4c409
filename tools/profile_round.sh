# Run on the GPU box: the evidence set of one round for the bs64 bench step.
#   usage: bash tools/profile_round.sh r03     (writes gpurun_out/<tag>_*; copy what is to be judged into profiles/)
TAG=${1:-rXX}
export TMPDIR=/tmp
rm -rf gpurun_out/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_$TAG.log 2>&1
python3 tools/step_breakdown.py gpurun_out/prof_$TAG > gpurun_out/${TAG}_bench_bs64_step_breakdown.txt 2>&1
cp $(ls gpurun_out/prof_$TAG/*/*_kernel_stats.csv | head -1) gpurun_out/${TAG}_bench_bs64_kernel_stats.csv
rm -rf gpurun_out/prof_$TAG
bash tools/bench_pmc.sh
cp gpurun_out/pmc_summary.csv gpurun_out/${TAG}_bench_bs64_pmc_summary.csv
cp gpurun_out/gemm128_hbm_traffic.json gpurun_out/${TAG}_gemm128_hbm_traffic.json
rm -rf gpurun_out/pmc
bash tools/prof_c2.sh $TAG > /dev/null 2>&1
for w in c2 c5 c5_hs512 defaults defaults_seg; do
  python3 bench.py --workload $w > gpurun_out/${TAG}_bench_$w.json 2> gpurun_out/${TAG}_bench_$w.err
done
python3 bench.py --workload c2 --batch 64 --no-cpu-baseline > gpurun_out/${TAG}_bench_c2_bs64.json 2> gpurun_out/${TAG}_bench_c2_bs64.err
python3 bench.py > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench_line.err
tail -3 gpurun_out/${TAG}_bench_line.err
head -12 gpurun_out/${TAG}_bench_bs64_step_breakdown.txt
