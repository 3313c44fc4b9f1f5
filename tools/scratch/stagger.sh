C=25165815
E=92274679
python tools/ab_step.py "chain=$C" "chain=$E" "chain=$C" "chain=$E" --batch 32 --rounds 8 2>&1 | tail -4
python tools/ab_step.py "chain=$C" "chain=$E" "chain=$C" "chain=$E" --prec f16 --rounds 8 2>&1 | tail -4
python tools/ab_step.py "chain=$C" "chain=$E" --prec f16x3 --rounds 8 2>&1 | tail -2
for o in "chain=$C" "chain=$E" "chain=$C" "chain=$E"; do MC_OPTS=$o python tools/control_bench.py s2g 32 2>&1 | tail -1 | cut -c1-120; done
for o in "chain=$C" "chain=$E" "chain=$C" "chain=$E"; do MC_OPTS=$o python tools/control_bench.py m2d 160 2>&1 | tail -1 | cut -c1-120; done
