#!/bin/bash
# round 3: hash-only top kernel variants: waves per SIMD x (sibling-leader bytes fetched with the item / just before its hash)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03t; mkdir -p $O
cd $R
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function"
for v in "3 0" "3 1" "4 0" "4 1"; do
  set -- $v
  touch stark-symphony_amd/csrc/ss_stwo.hip
  make -C stark-symphony_amd/csrc HIPFLAGS="$BASE -DSS_TOP_HASH_WAVES=$1 -DSS_TOP_YS_LATE=$2" > $O/make_$1_$2.log 2>&1 || { echo "make failed $v"; tail -5 $O/make_$1_$2.log; continue; }
  for w in stwo_2p20 stwo_2p16; do
  python bench.py --workload $w --steps 40 --warmup 4 --no-cpu-baseline --e2e 0 --distinct 16 > $O/bench_$1_$2_$w.json 2> $O/bench_$1_$2.err || { echo "bench failed $v"; tail -5 $O/bench_$1_$2.err; continue; }
  python - <<PY
import json
d=json.load(open('$O/bench_$1_$2_$w.json')); print("waves=$1 late=$2 $w", round(d['value']), round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernels_ms_per_step'].items() if k in ('stwo_merkle','stwo_top')}, round(d['alu_roofline']['frac'],4))
PY
  done
done
