#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof_cmd.sh <tag> <python script + args...>
# kernel-trace of an arbitrary tool -> gpurun_out/prof_<tag>/ + per-kernel summary (tools/rocpd_stats.py)
tag=$1; shift
root=$(pwd)
export TMPDIR=/tmp
mkdir -p $root/gpurun_out/prof_$tag
cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $root/gpurun_out/prof_$tag -o $tag -- python $root/"$@" > $root/gpurun_out/prof_$tag/run.log 2>&1
cd $root
tail -2 gpurun_out/prof_$tag/run.log
db=$(find gpurun_out/prof_$tag -name "*.db" | head -1)
python tools/rocpd_stats.py $db 16 | tee gpurun_out/prof_$tag/stats.txt
