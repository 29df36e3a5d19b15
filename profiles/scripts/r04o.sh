#!/bin/bash
# GPU call r04o: elections as tiles of candidates with per-voter thresholds (k_elections_tiled); early finalize of the last sub-batch
O=gpurun_out/r04o; mkdir -p $O
(timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_gpu_partition.py tests/test_gpu_baseline_configs.py tests/test_gpu_node.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log)
tail -5 $O/pytest.log
timeout 500 python profiles/knob_sweep.py 256 1000000 9 -- - SW_MID_PCT=0 SW_MID_PCT=80 SW_MID_PCT=94 SW_ELECT_CG=64 SW_ELECT_CG=256 SW_ELECT_IMPL=0 - > $O/knobs_256x1M.log 2>&1; cat $O/knobs_256x1M.log
GEN_MODE=2 GEN_P0=0.35 GEN_P1=0.02 timeout 300 python profiles/knob_sweep.py 256 1000000 5 -- - SW_ELECT_CG=64 SW_ELECT_CG=256 > $O/knobs_coin_256x1M.log 2>&1; cat $O/knobs_coin_256x1M.log
timeout 300 python profiles/knob_sweep.py 1024 2000000 3 -- - SW_MID_PCT=0 > $O/knobs_1024x2M.log 2>&1; cat $O/knobs_1024x2M.log
timeout 200 python bench.py --cpu-sample 0 --e2e-steps 0 > $O/bench_default.json 2> $O/bench_default.err; python - <<'PY'
import json
s=open('gpurun_out/r04o/bench_default.json').read()
d=json.loads(s[s.index('{"metric"'):])
print(d['value'], d['ms_per_step'], [(k['kernel'],k['avg_launch_us']) for k in d['roofline']['kernels']])
PY
