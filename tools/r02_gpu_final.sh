#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?"
python -c "
import json
d=json.load(open('$O/bench_default.json')); print(round(d['value']), d['ms_per_step'], d['roofline']['frac'], d['alu_roofline']['frac'], d['roofline']['traffic_source'])"
