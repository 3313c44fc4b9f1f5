#!/bin/bash
# LDS-conflict and TA/TCP-load counters per kernel over the serial fp32 B=64 step (and the f16 step)
root=$(pwd); export TMPDIR=/tmp; out=$root/gpurun_out/step_pmc; rm -rf $out; mkdir -p $out
SERIAL=92208535
cd /tmp
i=0
for set in "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  MC_CHAIN=$SERIAL timeout 300 rocprofv3 --kernel-trace --pmc $set -d $out/p$i -o s -- python $root/bench.py --no-cpu-baseline --no-extras --no-full-loop --steps 2 --warmup 1 > $out/p$i.log 2>&1
  db=$(find $out/p$i -name "*.db" | head -1)
  echo "== $set" >> $out/summary.txt
  python $root/tools/rocpd_pmc.py $db >> $out/summary.txt
done
