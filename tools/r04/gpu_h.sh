#!/bin/bash
# round 4: after the stark101 transcript rewrite -- full GPU suite, the stark101 bench lines, the parity fuzzer again
set -o pipefail
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1
echo "rc=$?" >> $O/gputest.log; tail -4 $O/gputest.log
grep -q "rc=0" $O/gputest.log || exit 1
python bench.py --workload stark101 --steps 1920 --warmup 6 --cpu-seconds 3 > $O/bench_stark101_4096.json 2> $O/bench_stark101.err; echo "stark101 rc=$?"
python bench.py --workload stark101 --proofs-per-gpu 8192 --steps 960 --warmup 6 --no-cpu-baseline --e2e 0 > $O/bench_stark101_8192.json 2>> $O/bench_stark101.err
python tools/fuzz_parity.py 20000 20261004 > $O/fuzz_parity.txt 2>&1; echo "fuzz rc=$?"; tail -1 $O/fuzz_parity.txt; head -2 $O/fuzz_parity.txt
python - <<'PY'
import json
for f in ("bench_stark101_4096","bench_stark101_8192"):
    d=json.loads([l for l in open("gpurun_out/r04/%s.json"%f) if l.startswith("{")][-1])
    print(f, round(d["value"]), round(d["ms_per_step"],4), "alu", round(d["alu_roofline"]["frac"],4), d["kernels_ms_per_step"])
PY
