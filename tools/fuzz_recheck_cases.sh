for spec in "405 8" "405 216" "405 286" "27 377"; do
  set -- $spec; seed=$1; idx=$2; n=$((idx+1))
  for mode in new old; do
    if [ $mode = old ]; then export TWOG_X3_ROWS128=0 TWOG_X3_XL=1 TWOG_DW_COLSUM=0 TWOG_PERSIST_GUARD=0; else unset TWOG_X3_ROWS128 TWOG_X3_XL TWOG_DW_COLSUM TWOG_PERSIST_GUARD; fi
    python3 tools/parity_fuzz.py $n $seed $idx > gpurun_out/fz_tmp.log 2>&1
    python3 - <<PY
import json
s=json.load(open('gpurun_out/parity_fuzz.json'))['summary']
print('seed $seed case $idx $mode:', 'passed', s['passed'], 'failed', s['failed'], 'accepted', [(a['candidate_owners'], a['confirmed_by_fp64'], round(a['worst_grad_rel_of_the_case'],5)) for a in s['cases_accepted_by_the_relu_signature_rule']], 'worst_grad_rel', round(s['worst_grad_rel'],6))
PY
  done
done
