#!/bin/bash
O=gpurun_out/$1; mkdir -p $O
python py-swirld_amd/build.py --force > $O/build.log 2>&1 || { echo BUILD FAILED; tail -5 $O/build.log; exit 1; }
SW_DEBUG_TIMING=1 timeout 100 python bench.py --cpu-sample 0 --e2e-steps 0 --steps 1 --warmup 0 --contexts 1 $2 > $O/dbg.json 2> $O/debug_timing.log; grep "dataflow\|sub-batch" $O/debug_timing.log | tail -8
