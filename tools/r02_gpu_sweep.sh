#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/sweep; mkdir -p $O
cd $R
timeout 2400 python tools/shape_sweep.py 300 20261003 > $O/sweep.txt 2>&1; echo "sweep rc=$?"
tail -4 $O/sweep.txt
