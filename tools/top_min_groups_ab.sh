#!/bin/bash
# kTopMinGroups A/B (csrc/ss_layout.h): a batch that gives fewer top-kernel groups than this is cut into smaller groups.
# 1 024 (default): 8 192 proofs -> groups of 8 proofs; 512: groups of 16 (fuller plans, 512 blocks for 768 slots); 2 048: groups of 4.
# build/ab/libss_A.so = -DSS_TOP_MIN_GROUPS=512, libss_B.so = 2048 (built by hand from the same tree).
cd ${GRAFT_REPO_ROOT:-$(pwd)}
cp stark-symphony_amd/libss_verify.so build/ab/libss_default.so
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_ms_per_step']
print('%10.0f proofs/s  %7.3f ms/step  merkle %.3f top %.3f  step/roof %.3f' % (d['value'], d['ms_per_step'], k.get('stwo_merkle',0), k.get('stwo_top',0), d['alu_roofline']['frac_of_step']))"; }
for rep in 1 2; do
for v in default A B; do
  cp build/ab/libss_$v.so stark-symphony_amd/libss_verify.so
  for n in 8192 16384; do
    echo -n "min_groups=$v n=$n alone:        "; python bench.py --proofs-per-gpu $n --steps 240 --warmup 6 --no-cpu-baseline --e2e 0 --distinct 16 2>/dev/null | line
    echo -n "min_groups=$v n=$n RCCL reduce:  "; SS_BENCH_GROUP_OF_ONE=1 python bench.py --proofs-per-gpu $n --steps 240 --warmup 6 --no-cpu-baseline --e2e 0 --distinct 16 2>/dev/null | line
  done
done
done
cp build/ab/libss_default.so stark-symphony_amd/libss_verify.so
