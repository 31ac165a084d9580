#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/h; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_prover.py tests/test_gpu_intermediates.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -15 $O/pytest.log
python tools/prover_bench.py 20 5 > $O/prover_bench.txt 2>&1; tail -4 $O/prover_bench.txt
python tools/prover_bench.py 20 3 blake2s > $O/prover_bench_b2s.txt 2>&1; tail -2 $O/prover_bench_b2s.txt
