#!/bin/bash
# round 3, run B: the GPU text reader
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03b; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_text.py -q -x > $O/pytest_text.log 2>&1; echo "pytest rc=$?" >> $O/pytest_text.log
tail -25 $O/pytest_text.log
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
tail -3 $O/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03b/bench.json')); print(round(d['value']), json.dumps(d.get('e2e'), indent=1))
PY
