#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/e; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "memoisation or baseline_configs or wrong_path" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
for w in stwo_2p20 stwo_2p16 stwo_wide256 stwo_fixture; do
  python bench.py --workload $w --steps 40 --warmup 4 --no-cpu-baseline --e2e 0 > $O/bench_$w.json 2> $O/bench_$w.err; echo "bench $w rc=$?"
done
python bench.py --workload stwo_2p20 --proofs-per-gpu 8192 --steps 100 --warmup 6 --no-cpu-baseline --e2e 0 > $O/bench_stwo_2p20_8192.json 2>> $O/bench_stwo_2p20.err
