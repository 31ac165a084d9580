#!/bin/bash
# full GPU suite + the bench lines that changed since pass G
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/k; mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?"
python bench.py --proofs-per-gpu 8192 --steps 200 --warmup 6 --no-cpu-baseline --e2e 0 > $O/bench_stwo_2p20_8192.json 2> $O/bench_8192.err
python bench.py --workload stark101 --steps 1920 --warmup 6 --cpu-seconds 3 > $O/bench_stark101_4096.json 2> $O/bench_stark101.err
python bench.py --workload stark101 --proofs-per-gpu 8192 --steps 960 --warmup 6 --no-cpu-baseline > $O/bench_stark101_8192.json 2>> $O/bench_stark101.err
python bench.py --workload stwo_2p16_blake2s --steps 60 --warmup 6 --no-cpu-baseline > $O/bench_stwo_2p16_blake2s.json 2> $O/bench_b2s.err
for f in $O/bench_*.json; do python -c "
import json
d=json.load(open('$f')); print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],3),'ms', round(d['alu_roofline']['frac'],3), {k:(round(v['proofs_per_s']),round(v['parse_MB_per_s_per_thread'])) for k,v in d.get('e2e',{}).items() if isinstance(v,dict)})"; done
