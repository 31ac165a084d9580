#!/bin/bash
# round 3, run A: GPU parity suite after the advisor fixes + strong-scaling bench; PCIe upload rate probe
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03a; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -6 $O/pytest.log
python bench.py --steps 30 --warmup 4 > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03a/bench_default.json')); print(round(d['value']), d['ms_per_step'], d['scaling'], d['roofline']['frac'], d['alu_roofline']['frac'], d.get('e2e'))
PY
python tools/r03/pcie_probe.py > $O/pcie_probe.txt 2>&1; cat $O/pcie_probe.txt
