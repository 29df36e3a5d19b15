#!/bin/bash
# GPU call r04ze: the one-runtime guard of HipRangeBackend under the suite's import order; window tests first
O=gpurun_out/r04ze; mkdir -p $O
(timeout 600 python -m pytest tests/test_gpu_window.py tests/test_gpu_strong_split.py tests/test_gpu_partition.py tests/test_gpu_order.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log); tail -3 $O/pytest.log
python - <<'PY'
import sys, importlib
sys.path.insert(0, '.')
import torch
L = importlib.import_module("py-swirld_amd._lib"); L.load(); print("torch first:", L.hip_runtime_paths())
PY
