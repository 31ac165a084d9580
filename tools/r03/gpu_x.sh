#!/bin/bash
# round 3: long differential fuzz at the final kernels (80 000 mutants per configuration and mode), then a 400-shape sweep
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03x; mkdir -p $O
cd $R
timeout 3000 python tools/fuzz_parity.py 80000 303909 > $O/fuzz_303909.txt 2>&1; echo "fuzz rc=$?"; tail -15 $O/fuzz_303909.txt
timeout 2400 python tools/shape_sweep.py 400 20261006 > $O/shape_sweep.txt 2>&1; echo "sweep rc=$?"; tail -2 $O/shape_sweep.txt
