#!/bin/bash
# GPU call r05f: chained round loops (SW_CHAIN) — tests first, then what the gaps were worth and which cut schedule suits it
O=gpurun_out/r05f; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_chain.py -m gpu -x -q > $O/pytest_chain.log 2>&1; tail -5 $O/pytest_chain.log
timeout 400 python profiles/knob_sweep.py 256 1000000 11 -- - SW_CHAIN=0 SW_CHAIN=1,SW_SHOT_EXTRA=4 SW_CHAIN=1,SW_SHOT_EXTRA=8 SW_PIPE=6 SW_PIPE=8 SW_PIPE=12 "SW_CUTS=0.015;0.05;0.12;0.25;0.45;0.7" "SW_CUTS=0.03;0.1;0.25;0.5;0.75" "SW_CUTS=0.008;0.03;0.0625;0.15;0.3;0.5;0.75" - SW_CHAIN=0 > $O/knobs_256x1M.log 2>&1; cat $O/knobs_256x1M.log
timeout 200 python profiles/knob_sweep.py 64 100000 9 -- - SW_CHAIN=0 SW_PIPE=8 - > $O/knobs_64x100k.log 2>&1; cat $O/knobs_64x100k.log
timeout 300 python profiles/knob_sweep.py 1024 2000000 3 -- - SW_CHAIN=0 > $O/knobs_1024x2M.log 2>&1; cat $O/knobs_1024x2M.log
timeout 200 python profiles/knob_sweep.py 256 10000000 3 -- - SW_CHAIN=0 SW_PIPE=8 > $O/knobs_256x10M.log 2>&1; cat $O/knobs_256x10M.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_gpu_errors.py tests/test_gpu_baseline_configs.py -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
