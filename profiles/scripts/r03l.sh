#!/bin/bash
# round 3: gallop as a default? (uniform shapes must not lose), the 10 M-event run, hot members with a wider band
O=gpurun_out/$1; mkdir -p $O
python py-swirld_amd/build.py --force > $O/build.log 2>&1 || { echo BUILD FAILED; tail -5 $O/build.log; exit 1; }
timeout 300 python profiles/knob_sweep.py 256 1000000 9 -- - SW_GALLOP=1 SW_GALLOP=2 SW_GALLOP=3 2>&1 | tee $O/knobs_gallop_256x1M.log
timeout 100 python profiles/knob_sweep.py 64 100000 9 -- - SW_GALLOP=2 2>&1 | tee $O/knobs_gallop_64x100k.log
GEN_MODE=2 GEN_P0=0.35 GEN_P1=0.02 timeout 200 python profiles/knob_sweep.py 256 1000000 3 -- - SW_GALLOP=2 SW_CHUNKS=1 2>&1 | tee $O/knobs_coin_stress.log
GEN_MODE=1 GEN_P0=0.02 timeout 200 python profiles/knob_sweep.py 256 1000000 3 -- - SW_GALLOP=2 SW_CHUNKS=1 2>&1 | tee $O/knobs_two_cliques.log
timeout 400 python bench.py --cpu-sample 0 --e2e-steps 0 --steps 3 --warmup 1 --contexts 1 --events 10000000 > $O/bench_256x10M.json 2> $O/bench_256x10M.err
python -c "
import json; d=json.load(open('$O/bench_256x10M.json')); print('256x10M', d['value'], d['ms_per_step'], d['roofline']['counters'])"
