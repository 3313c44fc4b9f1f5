#!/bin/bash
# usage (on the GPU box, from the repo root): tools/collect_profiles.sh <round tag, e.g. r02>
# Collects the rocprofv3 evidence bench.py's roofline numbers refer to, each in its own run (PMC passes never share a run
# with other trace domains): per-kernel time tables (default two-stream schedule and the serial single-stream schedule),
# MFMA-busy / clock counters (serial schedule), HBM FETCH_SIZE / WRITE_SIZE in two separate passes.
# Summaries land in gpurun_out/<tag>_*.txt; copy the ones to keep into profiles/.
tag=${1:-rXX}
root=$(pwd)
export TMPDIR=/tmp
out=$root/gpurun_out
mkdir -p $out
# single-stream schedule = the default chain mask (McOptions::chain, 763363319) without bits 5, 6, 9, 16 (two sample groups on two streams)
SERIAL=763297175
COMMIT=${GIT_COMMIT:-unknown}
BENCH="python $root/bench.py --no-cpu-baseline --no-extras --no-full-loop"
run() {   # run <name> <rocprof args...> -- <cmd...>
    name=$1; shift
    rm -rf $out/prof_$name; mkdir -p $out/prof_$name
    (cd /tmp && timeout 900 rocprofv3 "$@" > $out/prof_$name/run.log 2>&1)
    find $out/prof_$name -name "*.db" | head -1
}
db=$(run ${tag}_stats --kernel-trace -d $out/prof_${tag}_stats -o s -- $BENCH --steps 12 --warmup 3)
{ echo "# bench.py --steps 12 --warmup 3 (B=64): default schedule = two sample groups on two HIP streams (kernel intervals of the two groups overlap: sum of durations > wall time)"; python tools/rocpd_stats.py $db 24; } > $out/${tag}_kernel_stats_b64.txt
{ echo "# the same run: launch-by-launch timeline of the last ~1.3 steps (tools/rocpd_timeline.py; queue = HIP stream, gap < 0 = overlap with the previous kernel)"; python tools/rocpd_timeline.py $db 150; } > $out/${tag}_b64_timeline.txt
db=$(MC_CHAIN=$SERIAL run ${tag}_serial --kernel-trace -d $out/prof_${tag}_serial -o s -- $BENCH --steps 12 --warmup 3)
{ echo "# MC_CHAIN=$SERIAL bench.py --steps 12 --warmup 3 (B=64): single-stream schedule (whole-batch launches, no overlap) -- the clean per-kernel durations"; python tools/rocpd_stats.py $db 24; } > $out/${tag}_kernel_stats_b64_serial.txt
db=$(MC_CHAIN=$SERIAL run ${tag}_mfma --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY -d $out/prof_${tag}_mfma -o s -- $BENCH --steps 3 --warmup 1)
{ echo "# commit $COMMIT"; echo "# MC_CHAIN=$SERIAL (single-stream schedule) rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY"; echo "#   -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-full-loop   (B=64, T=196); summary by tools/rocpd_pmc.py"; echo "# MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (cycles * 1024 SIMDs), cycles = GRBM_GUI_ACTIVE / 8 XCDs; per-dispatch averages"; python tools/rocpd_pmc.py $db; } > $out/${tag}_pmc_mfma_busy.txt
dbf=$(run ${tag}_fetch --kernel-trace --pmc FETCH_SIZE -d $out/prof_${tag}_fetch -o s -- $BENCH --steps 4 --warmup 1)
dbw=$(run ${tag}_write --kernel-trace --pmc WRITE_SIZE -d $out/prof_${tag}_write -o s -- $BENCH --steps 4 --warmup 1)
{ echo "# commit $COMMIT"; echo "# rocprofv3 --kernel-trace --pmc FETCH_SIZE  and  --pmc WRITE_SIZE (separate passes) over"; echo "#   python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras --no-full-loop   (B=64, T=196; 5 steps + setup per pass); table by tools/hbm_traffic.py"; echo "# units: KiB per dispatch (average).  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 128-B requests of wide coalesced reads as 64 B -> fetch_x2 doubles it (calibrated on sampler_update_k in round 1); WRITE_SIZE exact."; python tools/hbm_traffic.py $dbf $dbw 5; } > $out/${tag}_pmc_hbm_traffic.txt
# FLOP ledger of one step of the serial schedule (what the launchers book; keyed kernel@grid like the trace) -> prices the roofline table
MC_CHAIN=$SERIAL GIT_COMMIT=$COMMIT python tools/flop_ledger.py 64 > $out/${tag}_flop_ledger_serial.txt 2> $out/${tag}_flop_ledger_serial.err
cp $out/${tag}_flop_ledger_serial.txt $root/profiles/ 2>/dev/null
# per-kernel roofline table from the two PMC summaries (bench.py reads its gemm_wp_k row as `roofline.dominant_kernel`)
cp $out/${tag}_pmc_mfma_busy.txt $out/${tag}_pmc_hbm_traffic.txt $root/profiles/ 2>/dev/null
python tools/kernel_roofline.py $tag > $out/${tag}_kernel_roofline.txt
tail -3 $out/${tag}_pmc_hbm_traffic.txt
head -12 $out/${tag}_kernel_stats_b64_serial.txt
