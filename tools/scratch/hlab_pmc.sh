#!/bin/bash
# PMC passes over tools/_bin/gemm_h_lab pmc (3 launches of gemm_hd_k<false> as in the library + 3 of the candidate)
root=$(pwd); export TMPDIR=/tmp; out=$root/gpurun_out/hlab_pmc; rm -rf $out; mkdir -p $out
cd /tmp
rocprofv3 --list-avail > $out/avail.txt 2>&1
i=0
for set in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $set -d $out/p$i -o p -f csv -- $root/tools/_bin/gemm_h_lab pmc > $out/p$i.log 2>&1
  f=$(find $out/p$i -name "*counter_collection.csv" | head -1)
  echo "== $set" >> $out/summary.txt
  python3 - "$f" >> $out/summary.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r['Kernel_Name'][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    print('  ', k, {c: round(sum(v) / len(v), 1) for c, v in d.items()})
PY
done
cat $out/summary.txt
