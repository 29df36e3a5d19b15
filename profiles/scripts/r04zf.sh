#!/bin/bash
# GPU call r04zf: the host-side edits after the last full suite (SW_PIPE bound, runtime guard): the tests that touch them
O=gpurun_out/r04zf; mkdir -p $O
(timeout 300 python -m pytest tests/test_gpu_errors.py tests/test_gpu_c_abi.py tests/test_gpu_node.py "tests/test_gpu_parity.py::test_pipelined_subbatches_match_oracle" "tests/test_gpu_parity.py::test_round_numbers_from_the_band_pass_or_from_the_rows" -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log); tail -3 $O/pytest.log
