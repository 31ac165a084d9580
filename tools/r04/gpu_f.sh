#!/bin/bash
# round 4: the transcript kernel around one inlined compression -- parity, then the shapes where the HEAD half shows
set -o pipefail
mkdir -p gpurun_out/r04g
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_intermediates.py tests/test_blake2s.py -m gpu -x -q > gpurun_out/r04g/tests.log 2>&1
echo "rc=$?" >> gpurun_out/r04g/tests.log; tail -5 gpurun_out/r04g/tests.log
grep -q "rc=0" gpurun_out/r04g/tests.log || exit 1
python bench.py --no-cpu-baseline --e2e 0 > gpurun_out/r04g/bench_default.json 2>gpurun_out/r04g/err.txt
python bench.py --proofs-per-gpu 8192 --steps 200 --warmup 6 --no-cpu-baseline --e2e 0 > gpurun_out/r04g/bench_8192.json 2>>gpurun_out/r04g/err.txt
python bench.py --workload stwo_wide256 --steps 60 --warmup 6 --no-cpu-baseline --e2e 0 > gpurun_out/r04g/bench_wide256.json 2>>gpurun_out/r04g/err.txt
python bench.py --workload stwo_2p16 --steps 60 --warmup 6 --no-cpu-baseline --e2e 0 > gpurun_out/r04g/bench_2p16.json 2>>gpurun_out/r04g/err.txt
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/r04g/bench_*.json')):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1])
    print(os.path.basename(f), round(d['value']), round(d['ms_per_step'],3), 'alu', round(d['alu_roofline']['frac'],4), {k:round(v,3) for k,v in d['kernels_ms_per_step'].items()})
PY
