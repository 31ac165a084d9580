#!/bin/bash
# round 3: memoisation depth once the byte compares are in the merkle kernel: T = log2(Q) + 1 / 2 / 3 (SHA-256; one less for Blake2s)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03w; mkdir -p $O
cd $R
for x in 3 1 2; do
  make -C stark-symphony_amd/csrc clean > /dev/null
  make -C stark-symphony_amd/csrc -j16 EXTRA="-DSS_TOP_EXTRA=$x" > $O/make_$x.log 2>&1 || { echo "make failed $x"; tail -5 $O/make_$x.log; continue; }
  for w in stwo_2p20 stwo_2p16 stwo_2p20_blake2s; do
  python bench.py --workload $w --steps 40 --warmup 4 --no-cpu-baseline --e2e 0 --distinct 16 > $O/bench_${x}_$w.json 2> $O/bench_$x.err || { echo "bench failed $x"; tail -5 $O/bench_$x.err; continue; }
  python - <<PY
import json
d=json.load(open('$O/bench_${x}_$w.json')); print("extra=$x $w", round(d['value']), round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernels_ms_per_step'].items() if k in ('stwo_merkle','stwo_top')}, d['config']['hash_compressions_executed_per_proof'])
PY
  done
done
