# The four cases the round-5 sweep accepted without a confirmed ReLU unit (profiles/r05c_*), one by one, verbose.
# usage (GPU box): bash tools/fuzz_recheck_cases.sh > gpurun_out/fuzz_recheck.txt 2>&1
export TWOG_FUZZ_VERBOSE=1
for spec in "405 8" "405 216" "405 286" "27 377"; do
  set -- $spec; seed=$1; idx=$2; n=$((idx+1))
  echo "== seed $seed case $idx"
  python3 tools/parity_fuzz.py $n $seed $idx 2>&1 | tail -12
  python3 - <<PY
import json
d=json.load(open('gpurun_out/parity_fuzz.json'))
s=d['summary']
print('seed $seed case $idx:', 'passed', s['passed'], 'failed', s['failed'], 'accepted', [(a['candidate_owners'], a['confirmed_by_fp64'], round(a['worst_grad_rel_of_the_case'],5)) for a in s['cases_accepted_by_the_relu_signature_rule']], 'worst_grad_rel', round(s['worst_grad_rel'],6))
for r in d['results']: print({k:r[k] for k in ('layers','att','mtype','gran','agg','training','given_seg','fp64_decisions_differing') if k in r})
PY
done
