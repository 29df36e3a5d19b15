#!/bin/bash
# round 3: a value equal to the column's frontier event is final — chunks on hashgraphs with silent members
O=gpurun_out/$1; mkdir -p $O
python py-swirld_amd/build.py --force > $O/build.log 2>&1 || { echo BUILD FAILED; tail -5 $O/build.log; exit 1; }
(timeout 700 python -m pytest tests/test_gpu_chunks.py tests/test_gpu_parity.py tests/test_gpu_random.py -m gpu -x -q > $O/pytest_subset.log 2>&1; echo "pytest rc=$?" >> $O/pytest_subset.log)
grep -E "^(FAILED|ERROR|[0-9]+ (passed|failed))|pytest rc|AssertionError" $O/pytest_subset.log | cut -c1-400 | tail -8
GEN_MODE=2 GEN_P0=0.95 GEN_P1=0.002 timeout 300 python profiles/knob_sweep.py 256 1000000 3 -- - SW_CHUNKS=1 SW_CHUNKS=8,SW_CHUNK_MIN=8192 SW_CHUNKS=8,SW_CHUNK_MIN=8192,SW_PIPE=2 SW_HALO=16384 2>&1 | tee $O/knobs_hot.log
GEN_MODE=2 GEN_P0=0.35 GEN_P1=0.02 timeout 200 python profiles/knob_sweep.py 256 1000000 3 -- - SW_CHUNKS=1 2>&1 | tee $O/knobs_coin_stress.log
GEN_MODE=3 GEN_P0=0.3 timeout 200 python profiles/knob_sweep.py 256 1000000 3 -- - SW_CHUNKS=1 2>&1 | tee $O/knobs_stale.log
timeout 100 python profiles/knob_sweep.py 256 1000000 5 -- - 2>&1 | tee $O/knobs_uniform.log
