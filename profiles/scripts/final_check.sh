#!/bin/bash
# translation-cache microbenchmark, full GPU suite, smoke, default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
python py-swirld_amd/build.py --force > /dev/null 2>&1
out=gpurun_out/${1:-final}; mkdir -p $out
hipcc --offload-arch=gfx950 -O2 profiles/microbench/vmm_remap.hip -o /tmp/vmm_remap 2> $out/vmm_remap_build.log
{ /tmp/vmm_remap 0 8 2; /tmp/vmm_remap 1 8 2; /tmp/vmm_remap 0 4 64; /tmp/vmm_remap 1 4 64; } > $out/vmm_remap.log 2>&1
grep "^mode" $out/vmm_remap.log
timeout 700 python -m pytest tests -m gpu -q -p no:cacheprovider > $out/pytest_gpu.log 2>&1; echo "rc=$?" >> $out/pytest_gpu.log
grep -E "Failed:|passed|failed|^rc=" $out/pytest_gpu.log | cut -c1-1500
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; tail -1 $out/smoke.log
timeout 400 python bench.py > $out/bench.json 2> $out/bench.err; cut -c1-400 $out/bench.json
