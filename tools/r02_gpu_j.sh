#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j; mkdir -p $O; rm -f $O/*.json
cd $R
for hq in 24 32 48; do for st in 16 24 32; do
  GPU_MAX_HW_QUEUES=$hq python bench.py --workload stark101 --steps 1920 --warmup 6 --no-cpu-baseline --graph streams --streams $st > $O/s101_4096_hq${hq}_st$st.json 2>> $O/err.txt
done; done
GPU_MAX_HW_QUEUES=32 python bench.py --workload stark101 --proofs-per-gpu 8192 --steps 960 --warmup 6 --no-cpu-baseline --graph streams --streams 24 > $O/s101_8192_hq32_st24.json 2>> $O/err.txt
GPU_MAX_HW_QUEUES=32 python bench.py --workload stwo_fixture --proofs-per-gpu 4096 --steps 480 --warmup 6 --no-cpu-baseline --e2e 0 --graph streams --streams 24 > $O/fixture_4096_hq32_st24.json 2>> $O/err.txt
for f in $O/*.json; do python -c "
import json,sys
d=json.load(open('$f')); print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step']*1e3,1),'us', d['config']['submission'][:22], round(d['alu_roofline']['frac'],3))"; done
