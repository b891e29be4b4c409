# PMC passes over one GEMM shape (GPU box): bash tools/gemm_pmc.sh M N K [akm bkm]
export TMPDIR=/tmp
rm -rf gpurun_out/gpmc
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
P2="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL"
P3="GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VALU_MFMA_COEXEC_CYCLES"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  rocprofv3 --pmc $P --output-format csv -d gpurun_out/gpmc/p$i -- python3 tools/gemm_pmc.py "$@" > gpurun_out/gpmc_$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
csv.field_size_limit(1 << 30)
tot = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob('gpurun_out/gpmc/p*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'gemm_' in r['Kernel_Name']:
            tot[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
for k in sorted(tot):
    print(f'{k:32s} {tot[k] / n[k]:16.0f} per dispatch ({n[k]} dispatches)')
PY
