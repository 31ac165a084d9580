#!/bin/bash
# round 3, run I: does overlapping consecutive Merkle launches at their edges help small per-GPU shares?
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03i; mkdir -p $O
cd $R
for ts in 1 2; do for n in 8192 65536; do
  st=$((n==8192 ? 200 : 40))
  python bench.py --proofs-per-gpu $n --steps $st --warmup 6 --no-cpu-baseline --e2e 0 --tail-streams $ts > $O/bench_${n}_ts$ts.json 2> $O/err_${n}_ts$ts.txt
  python - <<PY
import json
d=json.load(open('$O/bench_${n}_ts$ts.json')); print("n=$n tail_streams=$ts", round(d['value']), round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernels_ms_per_step'].items()}, 'alu', round(d['alu_roofline']['frac'],4))
PY
done; done
for w in stwo_wide256 stwo_2p16; do for ts in 1 2; do
  python bench.py --workload $w --steps 60 --warmup 6 --no-cpu-baseline --e2e 0 --tail-streams $ts > $O/bench_${w}_ts$ts.json 2> $O/err_${w}_ts$ts.txt
  python - <<PY
import json
d=json.load(open('$O/bench_${w}_ts$ts.json')); print("$w tail_streams=$ts", round(d['value']), round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernels_ms_per_step'].items()}, 'alu', round(d['alu_roofline']['frac'],4))
PY
done; done
