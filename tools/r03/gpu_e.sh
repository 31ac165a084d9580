#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03e; mkdir -p $O
cd $R
for nt in 1 0; do for th in 4 8 15; do
  echo "NT=$nt threads=$th"
  SS_STAGE_NT=$nt SS_STAGE_THREADS=$th python tools/e2e_bench.py --n 3072 --reps 4 2>>$O/e2e.err | python -c "
import json,sys
d=json.load(sys.stdin)
for k in ('json','wit'): print(k, round(d[k]['proofs_per_s_best']), round(d[k]['proofs_per_s_median']), round(d[k]['text_GB_per_s_best'],1), 'stage_ms', round(d[k]['stage_ms'],1), 'total', round(d[k]['total_ms_best'],1))"
done; done
