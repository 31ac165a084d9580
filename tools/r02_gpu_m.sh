#!/bin/bash
# after a top-kernel change: memoisation parity tests, a fuzz slice, the shape sweep, per-kernel times, bench
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/m; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
timeout 900 python tools/fuzz_parity.py 4000 5150515 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/fuzz.txt
timeout 900 python tools/shape_sweep.py 60 777 2>&1 | grep -v amdgpu.ids | tail -1 | tee $O/sweep.txt
python tools/top_probe.py 2>&1 | grep -v amdgpu.ids | tail -1 | tee $O/probe.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?"
python -c "
import json
d=json.load(open('$O/bench_default.json')); print(round(d['value']), d['ms_per_step'], d['roofline']['frac'], d['alu_roofline']['frac'])"
