#!/bin/bash
# round 3: adaptive window offset (per member: a fresh round's window starts a margin before the distance the previous round ended at)
O=gpurun_out/$1; mkdir -p $O
python py-swirld_amd/build.py --force > $O/build.log 2>&1 || { echo BUILD FAILED; tail -5 $O/build.log; exit 1; }
(timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_gpu_baseline_configs.py tests/test_gpu_node.py -m gpu -x -q > $O/pytest_adapt.log 2>&1; echo "pytest rc=$?" >> $O/pytest_adapt.log)
grep -E "^(FAILED|ERROR|[0-9]+ (passed|failed))|pytest rc|Error" $O/pytest_adapt.log | cut -c1-300 | tail -6
timeout 600 python profiles/knob_sweep.py 1024 2000000 3 -- - SW_SKIP_ADAPT=0 SW_SKIP_ADAPT=2 SW_SKIP_ADAPT=3 SW_SKIP_ADAPT=5 SW_SKIP_ADAPT=6 SW_SKIP_ADAPT=4,SW_TALLY_K=12 SW_SKIP_ADAPT=4,SW_TALLY_K=16 SW_SKIP_ADAPT=3,SW_TALLY_K=12 SW_SKIP_ADAPT=5,SW_TALLY_K=12 SW_SKIP_ADAPT=4,SW_TALLY_K=20 2>&1 | tee $O/knobs_1024x2M.log
GEN_MODE=2 GEN_P0=0.40 GEN_P1=0.02 timeout 600 python profiles/knob_sweep.py 1024 4000000 2 -- - SW_SKIP_ADAPT=0 SW_SKIP_ADAPT=4,SW_TALLY_K=12 SW_SKIP_ADAPT=4,SW_TALLY_K=16 2>&1 | tee $O/knobs_1024x4M_coin.log
GEN_MODE=1 GEN_P0=0.002 timeout 600 python profiles/knob_sweep.py 1024 4000000 2 -- - SW_SKIP_ADAPT=0 SW_SKIP_ADAPT=4,SW_TALLY_K=12 SW_SKIP_ADAPT=4,SW_TALLY_K=16 2>&1 | tee $O/knobs_1024x4M_cliques.log
timeout 300 python profiles/knob_sweep.py 512 1000000 3 -- - SW_SKIP_ADAPT=0 SW_SKIP_ADAPT=4,SW_TALLY_K=16 SW_SKIP_ADAPT=4,SW_TALLY_K=12 2>&1 | tee $O/knobs_512x1M.log
timeout 200 python profiles/knob_sweep.py 256 1000000 5 -- - SW_SKIP_ADAPT=4 SW_SKIP_ADAPT=6 SW_SKIP_ADAPT=8 2>&1 | tee $O/knobs_256x1M.log
