#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03l; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_prover.py -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -5 $O/pytest.log
timeout 900 python tools/prover_bench.py 20 3 sha256 1,2,3,4,6,8 48 > $O/prover_bench.txt 2>&1; grep -v amdgpu $O/prover_bench.txt | tail -12
