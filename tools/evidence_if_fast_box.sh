ms=$(python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")
echo "probe ms_per_step=$ms" > gpurun_out/evid_probe.log
ok=$(python3 -c "print(1 if float('$ms') < 65.0 else 0)")
if [ "$ok" = "1" ]; then
  bash tools/profile_round.sh r05 > gpurun_out/r05_profile_round.log 2>&1
  python3 bench.py --workload c2 --batch 64 > gpurun_out/r05_bench_c2_bs64.json 2> gpurun_out/r05_bench_c2_bs64.err
  echo "full set done" >> gpurun_out/evid_probe.log
fi
