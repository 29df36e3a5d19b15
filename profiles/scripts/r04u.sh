#!/bin/bash
# GPU call r04u: rewind as one launch, exhaustion marks moved by k_loop_init: parity tests + the default sweep
O=gpurun_out/r04u; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_gpu_errors.py tests/test_gpu_node.py tests/test_gpu_chunks.py tests/test_gpu_c_abi.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log); tail -3 $O/pytest.log
timeout 300 python profiles/knob_sweep.py 256 1000000 11 -- - - > $O/knobs_256x1M.log 2>&1; cat $O/knobs_256x1M.log
