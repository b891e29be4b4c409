# GPU box: calibrates the FETCH_SIZE counter on the GEMM kernels' own access patterns with launches whose true read
# traffic is known (one tile column / one tile: no operand is reused inside the launch; 5 launches, operands far larger
# than the 32 MB of L2):  NN 61440 x 128 x 512 reads A = 125.8 MB (+ 0.26 MB of B) with 64-byte row segments per k-tile;
# TT 128 x 128 x 245760 reads A and B = 125.8 MB each with 512-byte k rows.
export TMPDIR=/tmp
rm -rf gpurun_out/fcal
for SHAPE in "61440 128 512 0 0" "128 128 245760 1 1" "61440 128 2048 0 0"; do
  tag=$(echo $SHAPE | tr ' ' '_')
  # one counter per pass (FETCH_SIZE and WRITE_SIZE together exceed what the hardware collects in one pass: rocprofv3
  # aborts and then hangs in its signal handler -- hence the timeout as well)
  timeout 180 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/fcal/$tag -- python3 tools/gemm_pmc.py $SHAPE > gpurun_out/fcal_$tag.log 2>&1
done
python3 - <<PY
import csv, glob, collections
csv.field_size_limit(1 << 30)
for d in sorted(glob.glob('gpurun_out/fcal/*')):
    tot = collections.defaultdict(float); n = collections.Counter()
    for f in glob.glob(d + '/*/*_counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            if 'gemm_' in r['Kernel_Name'] or 'splitk' in r['Kernel_Name']:
                key = (r['Kernel_Name'].split('(')[0][-40:], r['Counter_Name'])
                tot[key] += float(r['Counter_Value']); n[key] += 1
    for k in sorted(tot):
        print(d.split('/')[-1], k[0], k[1], f'{tot[k] / n[k] * 1024 / 1e6:10.1f} MB per dispatch (raw counter x 1 KiB, {n[k]} dispatches)')
PY
