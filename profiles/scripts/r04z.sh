#!/bin/bash
# GPU call r04z: round numbers and sees-masks from the band pass (SW_FIN_BAND): parity, cost, how many events go back to their rows
O=gpurun_out/r04z; mkdir -p $O
(timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_gpu_chunks.py tests/test_gpu_window.py tests/test_gpu_node.py tests/test_gpu_strong_split.py tests/test_gpu_partition.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log); tail -4 $O/pytest.log
timeout 500 python profiles/knob_sweep.py 256 1000000 9 -- - SW_MID_PCT=0 SW_FIN_BAND=0 SW_MID_PCT=0,SW_FIN_BLOCKS=8192 - > $O/knobs_256x1M.log 2>&1; cat $O/knobs_256x1M.log
timeout 300 python profiles/fin_band_counts.py > $O/fin_band_counts.txt 2>&1; cat $O/fin_band_counts.txt
