#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof_step.sh <tag> [batch]
# kernel-trace of tools/step_bench.py -> gpurun_out/prof_<tag>/ + printed per-kernel summary
tag=${1:-x}; B=${2:-64}
root=$(pwd)
export TMPDIR=/tmp
mkdir -p $root/gpurun_out/prof_$tag
cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $root/gpurun_out/prof_$tag -o $tag -- python $root/tools/step_bench.py $B > $root/gpurun_out/prof_$tag/run.log 2>&1
cd $root
tail -1 gpurun_out/prof_$tag/run.log
db=$(find gpurun_out/prof_$tag -name "*.db" | head -1)
python tools/rocpd_stats.py $db 14
