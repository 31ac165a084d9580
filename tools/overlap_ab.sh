#!/bin/bash
# How the passes of the metric config are submitted: the HEAD/TAIL pipeline with one and two tail streams against
# whole passes on 2 / 3 / 4 independent streams, at 65 536 proofs per pass and at one GPU's 8 192-proof share.
cd ${GRAFT_REPO_ROOT:-$(pwd)}
line() { python3 -c "
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
k=d['kernels_ms_per_step']
print('%-44s %10.0f proofs/s  %8.3f ms/step  alu %.3f  merkle %.3f top %.3f' % (sys.argv[2], d['value'], d['ms_per_step'], d['alu_roofline']['frac'], k.get('stwo_merkle',0), k.get('stwo_top',0)))" "$@"; }
T=$(mktemp -d)
B="python bench.py --no-cpu-baseline --e2e 0 --distinct 16"
for rep in 1; do
for n in 65536 8192; do
  S=$([ $n = 65536 ] && echo 40 || echo 240)
  $B --proofs-per-gpu $n --steps $S --warmup 4 --tail-streams 1 > $T/o.json 2> $T/err && line $T/o.json "$n pipeline, 1 tail stream" || tail -3 $T/err
  $B --proofs-per-gpu $n --steps $S --warmup 4 --tail-streams 2 > $T/o.json 2> $T/err && line $T/o.json "$n pipeline, 2 tail streams" || tail -3 $T/err
  for s in 2 3 4; do
    $B --proofs-per-gpu $n --steps $S --warmup 4 --graph streams --streams $s > $T/o.json 2> $T/err && line $T/o.json "$n whole passes on $s streams" || tail -3 $T/err
  done
done
done
