#!/bin/bash
# round 3: lo rows handed over with the loop state; wide elections kernel for > 256 members
O=gpurun_out/$1; mkdir -p $O
python py-swirld_amd/build.py --force > $O/build.log 2>&1 || { echo BUILD FAILED; tail -5 $O/build.log; exit 1; }
(timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log)
grep -E "^(FAILED|ERROR|[0-9]+ (passed|failed))|pytest rc|Error" $O/pytest_gpu.log | cut -c1-300 | tail -8
timeout 300 python profiles/knob_sweep.py 256 1000000 9 -- - 2>&1 | tee $O/knobs_256x1M.log
timeout 100 python profiles/knob_sweep.py 64 100000 9 -- - 2>&1 | tee $O/knobs_64x100k.log
timeout 300 python profiles/knob_sweep.py 1024 2000000 3 -- - SW_ELECT_IMPL=0 SW_TALLY_K=12 2>&1 | tee $O/knobs_1024x2M.log
timeout 300 python profiles/knob_sweep.py 512 1000000 3 -- - SW_ELECT_IMPL=0 2>&1 | tee $O/knobs_512x1M.log
