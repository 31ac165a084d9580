#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
python tools/prover_bench.py 20 2 sha256 1,4,4,4,2,3,4 48 2>&1 | grep prove_many
timeout 600 python -m pytest tests/test_gpu_prover.py -q -x 2>&1 | tail -3
