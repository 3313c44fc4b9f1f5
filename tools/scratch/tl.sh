root=$(pwd); export TMPDIR=/tmp
rm -rf $root/gpurun_out/prof_tl; mkdir -p $root/gpurun_out/prof_tl
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $root/gpurun_out/prof_tl -o s -- python $root/bench.py --no-cpu-baseline --no-extras --no-full-loop --steps 12 --warmup 3 > $root/gpurun_out/prof_tl/run.log 2>&1)
db=$(find $root/gpurun_out/prof_tl -name "*.db" | head -1)
python tools/rocpd_timeline.py $db 150 > gpurun_out/r05_b64_timeline_bit26.txt
