#!/bin/bash
# GPU call r05zz: the last host-side tidy-up (one shot predictor for both loop paths): smoke, the parity file, the default line's value
O=gpurun_out/r05zz; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_node.py tests/test_gpu_errors.py -m gpu -x -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log
timeout 100 python profiles/knob_sweep.py 256 1000000 9 -- - SW_CHAIN=1 - > $O/knobs_256x1M.log 2>&1; cat $O/knobs_256x1M.log
