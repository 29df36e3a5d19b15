#!/bin/bash
# round 3: host-side knobs of the pass: cut patterns (short last sub-batch = short exposed finalize tail), first-shot slack
O=gpurun_out/$1; mkdir -p $O
python py-swirld_amd/build.py --force > $O/build.log 2>&1 || { echo BUILD FAILED; tail -5 $O/build.log; exit 1; }
timeout 500 python profiles/knob_sweep.py 256 1000000 9 -- - SW_SHOT_EXTRA=4 SW_SHOT_EXTRA=6 SW_SHOT_EXTRA=10 SW_SHOT_EXTRA=0 \
  "SW_CUTS=0.0625;0.3;0.55;0.8;0.95" "SW_CUTS=0.0625;0.35;0.65;0.9" "SW_CUTS=0.05;0.25;0.5;0.75;0.92;0.98" "SW_CUTS=0.03;0.1;0.3;0.55;0.8;0.95" "SW_CUTS=0.0625;0.5;0.94" \
  "SW_CUTS=0.0625;0.3;0.55;0.8;0.95,SW_SHOT_EXTRA=6" SW_BATCH=48 SW_BATCH=12 2>&1 | tee $O/knobs_256x1M.log
SW_DEBUG_TIMING=1 timeout 100 python bench.py --cpu-sample 0 --e2e-steps 0 --steps 3 --warmup 1 --contexts 1 > /dev/null 2> $O/debug_timing.log; grep "sub-batch\|stages" $O/debug_timing.log | tail -6
