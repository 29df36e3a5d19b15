#!/bin/bash
# GPU call r04s: torch.cuda initialised after the windowed-table tests (file order window -> strong_split)
O=gpurun_out/r04s; mkdir -p $O
(timeout 600 python -m pytest tests/test_gpu_window.py tests/test_gpu_strong_split.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log); tail -3 $O/pytest.log
timeout 300 python profiles/knob_sweep.py 256 1000000 9 -- - SW_PIPE=4 SW_PIPE=3 SW_PIPE=6 - > $O/knobs_pipe.log 2>&1; cat $O/knobs_pipe.log
timeout 300 python profiles/knob_sweep.py 256 10000000 3 -- - SW_PIPE=4 SW_PIPE=6 SW_PIPE=8 > $O/knobs_pipe_10M.log 2>&1; cat $O/knobs_pipe_10M.log
