#!/bin/bash
# round 3, run F: full GPU suite + default bench after the GPU text reader
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03f; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -6 $O/pytest.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?"; tail -2 $O/bench_default.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03f/bench_default.json')); print(round(d['value']), d['ms_per_step'], d['scaling'], d['roofline']['frac'], d['alu_roofline']['frac']); print(json.dumps(d.get('e2e'),indent=1))
PY
