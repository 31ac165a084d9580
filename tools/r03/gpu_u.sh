#!/bin/bash
# round 3: hash-only top kernel reads the query kernel's plan; parity + timing
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03u; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_intermediates.py -x -q -m gpu > $O/parity_on.txt 2>&1; echo "parity rc=$?"; tail -3 $O/parity_on.txt
for w in stwo_2p20 stwo_2p16 stwo_2p20_blake2s stwo_wide256; do
python bench.py --workload $w --steps 60 --warmup 6 --no-cpu-baseline --e2e 0 --distinct 16 > $O/bench_$w.json 2> $O/bench.err
done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/r03u/bench_*.json')):
    try: d=json.load(open(f))
    except Exception as e: print(f,'BAD',e); continue
    print(os.path.basename(f), round(d['value']), round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernels_ms_per_step'].items()})
PY
