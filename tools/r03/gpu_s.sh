#!/bin/bash
# round 3: byte compares of the pair memoisation in the merkle kernel (lay.mchk): parity both ways, then A/B timing
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03s; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_intermediates.py -x -q -m gpu > $O/parity_on.txt 2>&1; echo "parity (merkle checks) rc=$?"; tail -3 $O/parity_on.txt
SS_MERKLE_CHECKS=0 timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/parity_off.txt 2>&1; echo "parity (top checks) rc=$?"; tail -3 $O/parity_off.txt
timeout 900 python tools/fuzz_parity.py 3000 303777 > $O/fuzz.txt 2>&1; echo "fuzz rc=$?"; tail -4 $O/fuzz.txt
for i in 1 2; do
python bench.py --steps 60 --warmup 6 --no-cpu-baseline --e2e 0 --distinct 16 > $O/bench_on_$i.json 2> $O/bench_on.err
SS_MERKLE_CHECKS=0 python bench.py --steps 60 --warmup 6 --no-cpu-baseline --e2e 0 --distinct 16 > $O/bench_off_$i.json 2> $O/bench_off.err
done
python bench.py --workload stwo_2p16 --steps 60 --warmup 6 --no-cpu-baseline --e2e 0 > $O/bench_2p16_on.json 2>> $O/bench_on.err
SS_MERKLE_CHECKS=0 python bench.py --workload stwo_2p16 --steps 60 --warmup 6 --no-cpu-baseline --e2e 0 > $O/bench_2p16_off.json 2>> $O/bench_off.err
python bench.py --workload stwo_2p20_blake2s --steps 60 --warmup 6 --no-cpu-baseline --e2e 0 > $O/bench_b2s_on.json 2>> $O/bench_on.err
SS_MERKLE_CHECKS=0 python bench.py --workload stwo_2p20_blake2s --steps 60 --warmup 6 --no-cpu-baseline --e2e 0 > $O/bench_b2s_off.json 2>> $O/bench_off.err
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/r03s/bench_*.json')):
    try: d=json.load(open(f))
    except Exception as e: print(f,'BAD',e); continue
    print(os.path.basename(f), round(d['value']), round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernels_ms_per_step'].items()})
PY
