#!/bin/bash
# round 4: full -m gpu suite + default bench after the device guards / e2e changes
set -o pipefail
mkdir -p gpurun_out/r04e
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r04e/gputest.log 2>&1
echo "rc=$?" >> gpurun_out/r04e/gputest.log
tail -25 gpurun_out/r04e/gputest.log
python bench.py > gpurun_out/r04e/bench.json 2> gpurun_out/r04e/bench.err; tail -c 300 gpurun_out/r04e/bench.err
python -c "
import json
d=json.loads([l for l in open('gpurun_out/r04e/bench.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['alu_roofline']['frac'])
for k,v in d['e2e'].items():
    if isinstance(v,dict): print(k, round(v['proofs_per_s']), v.get('text_GB_per_s') or v.get('link_GB_per_s'))
"
