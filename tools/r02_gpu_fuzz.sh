#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/fuzz; mkdir -p $O
cd $R
timeout 2400 python tools/fuzz_parity.py 40000 777003 > $O/fuzz_40000c.txt 2>&1; echo "fuzz rc=$?"
tail -4 $O/fuzz_40000c.txt
