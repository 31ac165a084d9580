#!/bin/bash
# round 3: differential fuzz of the kernels at HEAD (per-proof top-sibling layout, cold kernel) against the oracle
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03fuzz; mkdir -p $O
cd $R
timeout 2400 python tools/fuzz_parity.py ${1:-20000} ${2:-303001} > $O/fuzz_$2.txt 2>&1; echo "fuzz rc=$?"
tail -4 $O/fuzz_$2.txt
timeout 1200 python tools/shape_sweep.py 150 20261004 > $O/shape_sweep.txt 2>&1; echo "sweep rc=$?"; tail -3 $O/shape_sweep.txt
