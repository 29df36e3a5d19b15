#!/bin/bash
O=gpurun_out/$1; mkdir -p $O
python py-swirld_amd/build.py --force > $O/build.log 2>&1 || { echo BUILD FAILED; tail -5 $O/build.log; exit 1; }
(SW_TALLY_PF=1 timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "256 or 64 or 16" > $O/pytest_pf.log 2>&1; echo "pytest rc=$?" >> $O/pytest_pf.log)
grep -E "^(FAILED|ERROR|[0-9]+ (passed|failed))|pytest rc" $O/pytest_pf.log | tail -3
timeout 200 python profiles/knob_sweep.py 256 1000000 9 -- - SW_TALLY_PF=1 - SW_TALLY_PF=1 2>&1 | tee $O/knobs_256x1M.log
timeout 100 python profiles/knob_sweep.py 64 100000 9 -- - SW_TALLY_PF=1 2>&1 | tee $O/knobs_64x100k.log
