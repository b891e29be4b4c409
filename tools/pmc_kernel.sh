#!/bin/bash
# hardware counters per kernel name for a command:  bash tools/pmc_kernel.sh "<counters>" <name filter> -- python3 prog.py ...
CTRS="$1"; FILT="$2"; shift 3
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp; rm -rf $R/gpurun_out/pmck
rocprofv3 --pmc $CTRS --output-format csv -d $R/gpurun_out/pmck -- "$@" > /dev/null 2>&1
cd $R; python3 - "$FILT" <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob('gpurun_out/pmck/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if sys.argv[1] in r['Kernel_Name']:
            k = r['Kernel_Name'][:60]; agg[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
for k, v in agg.items():
    print(k, {c: round(x / n[(k, c)]) for c, x in v.items()}, 'dispatches', max(n[(k, c)] for c in v))
PY
rm -rf gpurun_out/pmck
