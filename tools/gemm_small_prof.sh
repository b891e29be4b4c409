# run on the GPU box: kernel-only durations of the 64x64-tile GEMM class over (tiles, K)
export TMPDIR=/tmp
rm -rf gpurun_out/gs
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gs -- python3 tools/gemm_small_prof.py > gpurun_out/gemm_small.log 2>&1
python3 - <<PY
import csv, glob, json
cfg = json.load(open('gpurun_out/gemm_small_cfg.json'))
rows = [r for r in csv.DictReader(open(glob.glob('gpurun_out/gs/*/*_kernel_trace.csv')[0])) if 'gemm_kernel' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
reps = cfg['reps']
for i, c in enumerate(cfg['configs']):
    d = sorted((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows[i * reps:(i + 1) * reps])
    fl = 2.0 * c['groups'] * c['M'] * c['N'] * c['K']
    med = d[len(d) // 2]
    print(f"groups={c['groups']} M={c['M']} N={c['N']} K={c['K']} bkm={c['bkm']} tiles={c['tiles']:5d} grid={rows[i*reps]['Grid_Size_X']} med_us={med:7.1f} min_us={d[0]:7.1f} TF={fl/med/1e6:6.1f}")
PY
