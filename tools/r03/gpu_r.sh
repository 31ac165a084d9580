#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
python tools/e2e_bench.py --n 4096 --reps 4 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin)
print({k:(round(v['proofs_per_s_best']),round(v['proofs_per_s_median']),round(v['text_GB_per_s_best'],1),round(v['stage_ms'],1)) for k,v in d.items() if isinstance(v,dict)})"
for th in 4 12 16; do SS_STAGE_THREADS=$th python tools/e2e_bench.py --n 4096 --reps 3 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin)
print('threads $th', {k:(round(v['proofs_per_s_best']),round(v['text_GB_per_s_best'],1),round(v['stage_ms'],1)) for k,v in d.items() if isinstance(v,dict)})"; done
