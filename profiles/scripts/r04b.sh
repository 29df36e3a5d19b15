#!/bin/bash
# GPU call r04b: new GPU tests of this round (event-range split, bulk find_order, knob validation, table-size fix,
# pruned sweep variants), then the bench line with find_order stage clocks and the emulated split
O=gpurun_out/r04b; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_strong_split.py tests/test_gpu_order.py tests/test_gpu_errors.py tests/test_gpu_window.py \
   "tests/test_gpu_parity.py::test_kernel_variants_agree" tests/test_gpu_random.py -m gpu -x -q > $O/pytest_new.log 2>&1; echo "pytest rc=$?" >> $O/pytest_new.log)
tail -15 $O/pytest_new.log
SW_DEBUG_TIMING=1 timeout 300 python bench.py --cpu-sample 0 --e2e-steps 0 --steps 5 --warmup 1 --emulate-parts 4 > $O/bench_default.json 2> $O/bench_default.err
grep "find_order\]" $O/bench_default.err | tail -8
python - $O <<'PY'
import json,sys
d=json.load(open(sys.argv[1]+"/bench_default.json"))
print("value %.1f M ev/s  %.3f ms/step  find_order %.2f ms  with_order %.1f M" % (d["value"]/1e6, d["ms_per_step"], d["find_order_ms"], (d["value_with_order"] or 0)/1e6))
print("strong:", json.dumps(d["strong"]))
PY
SW_ORDER_BULK=0 SW_DEBUG_TIMING=1 timeout 300 python bench.py --cpu-sample 0 --e2e-steps 0 --steps 2 --warmup 1 > $O/bench_order_old.json 2> $O/bench_order_old.err
grep "find_order\]" $O/bench_order_old.err | tail -5
