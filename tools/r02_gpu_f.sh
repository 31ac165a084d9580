#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/f; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
python bench.py --workload stark101 --steps 96 --warmup 6 --cpu-seconds 3 > $O/bench_stark101.json 2> $O/bench_stark101.err; echo "s101 rc=$?"
python bench.py --workload stark101 --steps 96 --warmup 6 --no-cpu-baseline --graph off > $O/bench_stark101_eager.json 2>> $O/bench_stark101.err
python bench.py --workload stark101 --proofs-per-gpu 8192 --steps 96 --warmup 6 --no-cpu-baseline > $O/bench_stark101_8192.json 2>> $O/bench_stark101.err
python bench.py --workload stwo_wide256 --steps 40 --warmup 4 --no-cpu-baseline --e2e 0 > $O/bench_wide.json 2> $O/bench_wide.err
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?"
