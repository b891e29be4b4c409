# Run on the GPU box: hardware-counter passes over one bench step (separate rocprofv3 --pmc runs, no tracing domains).
#   pass f: FETCH_SIZE   pass w: WRITE_SIZE   pass s: SQ / GRBM cycle counters
# Summaries: python3 tools/pmc_summary.py gpurun_out/pmc -> profiles/
export TMPDIR=/tmp
rm -rf gpurun_out/pmc
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc/f -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc/w -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_w.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc/s -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_s.log 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc gpurun_out/pmc_summary.csv gpurun_out/gemm128_hbm_traffic.json gpurun_out/gemm128_traffic_by_launch.csv
