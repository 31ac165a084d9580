#!/bin/bash
set -o pipefail
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "stark101 or s101 or fuzz" > $O/s101_tests.log 2>&1; echo "rc=$?" >> $O/s101_tests.log; tail -3 $O/s101_tests.log
grep -q "rc=0" $O/s101_tests.log || exit 1
for i in 1 2; do
python bench.py --workload stark101 --steps 1920 --warmup 6 --cpu-seconds 3 > $O/bench_stark101_4096.json 2> $O/bench_stark101.err
python bench.py --workload stark101 --proofs-per-gpu 8192 --steps 960 --warmup 6 --no-cpu-baseline --e2e 0 > $O/bench_stark101_8192.json 2>> $O/bench_stark101.err
python - <<'PY'
import json
for f in ("bench_stark101_4096","bench_stark101_8192"):
    d=json.loads([l for l in open("gpurun_out/r04/%s.json"%f) if l.startswith("{")][-1])
    print(f, round(d["value"]), round(d["ms_per_step"],4), "alu", round(d["alu_roofline"]["frac"],4), {k:round(v,4) for k,v in d["kernels_ms_per_step"].items()})
PY
done
python tools/fuzz_parity.py 20000 20261004 > $O/fuzz_parity.txt 2>&1; echo "fuzz rc=$?"; head -2 $O/fuzz_parity.txt | tail -1
