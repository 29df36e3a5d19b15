#!/bin/bash
# GPU call r04za: non-temporal stores for the round numbers / sees-masks the band pass writes (end-of-kernel write-back)
O=gpurun_out/r04za; mkdir -p $O
timeout 500 python profiles/knob_sweep.py 256 1000000 11 -- - SW_FIN_NT=0 - SW_FIN_NT=0 SW_FIN_BAND=0 > $O/knobs_256x1M.log 2>&1; cat $O/knobs_256x1M.log
timeout 300 python profiles/knob_sweep.py 1024 2000000 3 -- - SW_FIN_NT=0 > $O/knobs_1024x2M.log 2>&1; cat $O/knobs_1024x2M.log
SW_DEBUG_CLOCKS=2 timeout 120 python profiles/resolve_time.py 256 1000000 2>&1 | tail -3
SW_FIN_NT=0 SW_DEBUG_CLOCKS=2 timeout 120 python profiles/resolve_time.py 256 1000000 2>&1 | tail -3
(timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log); tail -3 $O/pytest.log
