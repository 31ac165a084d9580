#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03m; mkdir -p $O
cd $R
python tools/e2e_bench.py --n 4096 --reps 4 --files > $O/e2e_4096_files.json 2> $O/e2e.err; python -c "
import json; d=json.load(open('$O/e2e_4096_files.json'))
print({k:(round(v['proofs_per_s_best']),round(v.get('files_proofs_per_s_best',0))) for k,v in d.items() if isinstance(v,dict)})"
python tools/prover_bench.py 20 2 sha256 1,4,4,2 48 2>&1 | grep prove_many
timeout 600 python -m pytest tests/test_gpu_text.py -q -x 2>&1 | tail -3
