#!/bin/bash
# GPU call r05z: the rebuilt library (an unused host variable removed): smoke + the parity file
O=gpurun_out/r05z; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_chain.py -m gpu -x -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log
