#!/bin/bash
# round 3, run O: top-kernel group size for one GPU's 8 192-proof share (and 16 384 / 32 768: N = 4, 2)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03o; mkdir -p $O
cd $R
for n in 8192 16384 32768; do for mg in 1024 1536 2048 3072 4096; do
  SS_TOP_MIN_GROUPS=$mg python bench.py --proofs-per-gpu $n --steps 150 --warmup 6 --no-cpu-baseline --e2e 0 > $O/b_${n}_$mg.json 2> $O/err.txt
  python - <<PY
import json
d=json.load(open('$O/b_${n}_$mg.json')); print("n=$n min_groups=$mg", round(d['value']), round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernels_ms_per_step'].items() if k in ('stwo_merkle','stwo_top')})
PY
done; done
