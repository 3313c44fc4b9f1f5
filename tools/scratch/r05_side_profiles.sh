#!/bin/bash
# round-5 refresh of the side-config kernel tables (two-stream + serial) and the reduced-precision serial tables
root=$(pwd); export TMPDIR=/tmp
SERIAL=25099671
{ echo "# tools/prof_cmd.sh s2g tools/control_bench.py s2g 32: BASELINE configs[2] per-GPU share (L=128, 8 + 2 layers, 32 x 196 frames, raw-audio condition), 2 x 50-step DDIM";
  bash tools/prof_cmd.sh s2g tools/control_bench.py s2g 32 | tail -18;
  echo; echo "# the same, single-stream schedule (MC_OPTS=chain=$SERIAL)";
  MC_OPTS=chain=$SERIAL bash tools/prof_cmd.sh s2gs tools/control_bench.py s2g 32 | tail -18; } > gpurun_out/r05_kernel_stats_s2g.txt 2>&1
{ echo "# tools/prof_cmd.sh m2d tools/control_bench.py m2d 160: BASELINE configs[3] per-GPU share (L=64, 4 + 3 layers, 160 windows x 120 frames), 2 x 50-step DDIM";
  bash tools/prof_cmd.sh m2d tools/control_bench.py m2d 160 | tail -18;
  echo; echo "# the same, single-stream schedule (MC_OPTS=chain=$SERIAL)";
  MC_OPTS=chain=$SERIAL bash tools/prof_cmd.sh m2ds tools/control_bench.py m2d 160 | tail -18; } > gpurun_out/r05_kernel_stats_m2d.txt 2>&1
for prec in f16 f16x3; do
  { echo "# tools/prof_cmd.sh ${prec}s tools/ab_step.py chain=$SERIAL --prec $prec --rounds 2 --steps 6: B=64 headline shape, single-stream schedule, precision $prec";
    bash tools/prof_cmd.sh ${prec}s tools/ab_step.py "chain=$SERIAL" --prec $prec --rounds 2 --steps 6 | tail -18; } > gpurun_out/r05_kernel_stats_b64_${prec}_serial.txt 2>&1
done
for i in 1 2; do python bench.py --no-extras --no-cpu-baseline --no-full-loop --steps 100 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench lease run', d['value'], d['ms_per_step'], d['roofline']['frac'])"; done
