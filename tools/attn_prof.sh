# rocprofv3 kernel-trace of the segment-level and frame-level attention calls (tools/attn_bench.py); run on the GPU box
export TMPDIR=/tmp
rm -rf gpurun_out/ap
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ap -- python3 tools/attn_bench.py > gpurun_out/attn_bench.log 2>&1
python3 - <<PY
import csv, glob
rows = list(csv.DictReader(open(glob.glob("gpurun_out/ap/*/*_kernel_stats.csv")[0])))
for r in rows:
    if "attn" in r["Name"]:
        print(r["Name"][:60], "calls", r["Calls"], "avg_us", float(r["AverageNs"]) / 1e3, "min_us", float(r["MinNs"]) / 1e3, "max_us", float(r["MaxNs"]) / 1e3)
PY
