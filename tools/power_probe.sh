#!/bin/bash
# sample clocks / power while the B=64 step runs (the complete 1000-step loop is ~20 s of steady load)
mkdir -p gpurun_out/r3g
(python bench.py --no-cpu-baseline --no-extras --steps 100 --warmup 5 > gpurun_out/r3g/bench.json 2>/dev/null) &
BP=$!
: > gpurun_out/r3g/smi.txt
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "sclk|fclk|Power|junction" | sed -e 's/.*: //' | tr '\n' ' ' >> gpurun_out/r3g/smi.txt; echo >> gpurun_out/r3g/smi.txt
  sleep 0.7
done
grep -v "(9[0-9]Mhz)\|(1[0-9][0-9]Mhz)" gpurun_out/r3g/smi.txt | tail -40
