#!/bin/bash
# GPU call r05h: one fill kernel in front of the elections, the last loop's read-back beside the tail, write-through row stores of the
# sweep (SW_CHUNK_CFG=3), shot slack
O=gpurun_out/r05h; mkdir -p $O
timeout 400 python profiles/knob_sweep.py 256 1000000 11 -- - SW_CHUNK_CFG=3 SW_SHOT_EXTRA=0 SW_SHOT_EXTRA=1 SW_CHAIN=0 - > $O/knobs_256x1M.log 2>&1; cat $O/knobs_256x1M.log
timeout 600 python -m pytest tests/test_gpu_chain.py tests/test_gpu_parity.py tests/test_gpu_partition.py tests/test_gpu_node.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
