#!/bin/bash
# round 3: tests for > 256 members (wide elections, partitioned), window offset x window size grids
O=gpurun_out/$1; mkdir -p $O
python py-swirld_amd/build.py --force > $O/build.log 2>&1 || { echo BUILD FAILED; tail -5 $O/build.log; exit 1; }
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_partition.py -m gpu -x -q > $O/pytest_wide.log 2>&1; echo "pytest rc=$?" >> $O/pytest_wide.log)
grep -E "^(FAILED|ERROR|[0-9]+ (passed|failed))|pytest rc|Error" $O/pytest_wide.log | cut -c1-300 | tail -6
timeout 600 python profiles/knob_sweep.py 1024 2000000 3 -- - SW_SKIP=4,SW_TALLY_K=20 SW_SKIP=6,SW_TALLY_K=16 SW_SKIP=8,SW_TALLY_K=16 SW_SKIP=8,SW_TALLY_K=12 SW_SKIP=10,SW_TALLY_K=12 SW_SKIP=6,SW_TALLY_K=20 SW_SKIP=4,SW_TALLY_K=24 2>&1 | tee $O/knobs_1024x2M.log
timeout 300 python profiles/knob_sweep.py 256 1000000 7 -- - SW_SKIP=4,SW_TALLY_K=24 SW_SKIP=6,SW_TALLY_K=22 SW_SKIP=6,SW_TALLY_K=20 SW_SKIP=8,SW_TALLY_K=20 SW_SKIP=4,SW_TALLY_K=28 SW_SKIP=2,SW_TALLY_K=28 2>&1 | tee $O/knobs_256x1M.log
