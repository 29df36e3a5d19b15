#!/bin/bash
# GPU call r04r: tests of the new knobs; sub-batch cut schedule revisited with the throttled finalize
O=gpurun_out/r04r; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_errors.py tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py tests/test_gpu_window.py tests/test_gpu_strong_split.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log)
tail -4 $O/pytest.log
timeout 800 python profiles/knob_sweep.py 256 1000000 9 -- - SW_CUTS=0.03:0.2:0.4:0.6:0.8 SW_CUTS=0.0625:0.3:0.55:0.8 SW_CUTS=0.0625:0.2:0.35:0.5:0.65:0.8 SW_CUTS=0.1:0.3:0.5:0.7:0.85 SW_CUTS=0.0625:0.25:0.4375:0.625:0.8125:0.92 SW_CUTS=0.04:0.14:0.3:0.5:0.7:0.87 - > $O/knobs_cuts_256x1M.log 2>&1; cat $O/knobs_cuts_256x1M.log
