#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03k; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_text.py tests/test_gpu_parity.py -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -25 $O/pytest.log
