#!/bin/bash
# round 3: full GPU suite + smoke + differential fuzz + shape sweep at the kernels with the byte compares in the merkle kernel
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03v; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.txt
timeout 2400 python tools/fuzz_parity.py 20000 303303 > $O/fuzz_303303.txt 2>&1; echo "fuzz rc=$?"; tail -12 $O/fuzz_303303.txt
timeout 1500 python tools/shape_sweep.py 150 20261005 > $O/shape_sweep.txt 2>&1; echo "sweep rc=$?"; tail -3 $O/shape_sweep.txt
