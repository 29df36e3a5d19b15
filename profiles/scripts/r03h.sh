#!/bin/bash
# round 3: verdicts of the tally reduced per workgroup before the atomics — parity, bench, phases
O=gpurun_out/$1; mkdir -p $O
python py-swirld_amd/build.py --force > $O/build.log 2>&1 || { echo BUILD FAILED; tail -5 $O/build.log; exit 1; }
(timeout 700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_gpu_chunks.py tests/test_gpu_node.py -m gpu -x -q > $O/pytest_subset.log 2>&1; echo "pytest rc=$?" >> $O/pytest_subset.log)
grep -E "^(FAILED|ERROR|[0-9]+ (passed|failed))|pytest rc" $O/pytest_subset.log | cut -c1-300 | tail -8
timeout 300 python profiles/knob_sweep.py 256 1000000 9 -- - SW_TALLY_K=32 SW_TALLY_K=24 SW_PIPE=1 2>&1 | tee $O/knobs_256x1M.log
timeout 200 python profiles/knob_sweep.py 64 100000 9 -- - 2>&1 | tee $O/knobs_64x100k.log
timeout 300 python profiles/knob_sweep.py 1024 2000000 3 -- - SW_TALLY_K=12 2>&1 | tee $O/knobs_1024x2M.log
GEN_MODE=2 GEN_P0=0.95 GEN_P1=0.002 timeout 300 python profiles/knob_sweep.py 256 1000000 3 -- - SW_GALLOP=1 2>&1 | tee $O/knobs_hot.log
SW_DEBUG_CLOCKS=1 SW_PIPE=1 timeout 120 python profiles/loop_phases.py 256 1000000 > $O/loop_phases_256.txt 2>&1
SW_DEBUG_CLOCKS=1 SW_PIPE=1 timeout 120 python profiles/loop_phases.py 64 100000 > $O/loop_phases_64.txt 2>&1
grep -E "iteration period|end - entry|end -> next|resolve end|compared" $O/loop_phases_256.txt $O/loop_phases_64.txt
