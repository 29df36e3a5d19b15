#!/bin/bash
# GPU call r04t: find_order with the narrowed bracket of k_order_bounds: tests, host laps
O=gpurun_out/r04t; mkdir -p $O
(timeout 600 python -m pytest tests/test_gpu_order.py tests/test_gpu_window.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log); tail -3 $O/pytest.log
timeout 200 python profiles/order_laps.py 256 1000000 > $O/order_laps_256x1M.txt 2>&1; cat $O/order_laps_256x1M.txt
