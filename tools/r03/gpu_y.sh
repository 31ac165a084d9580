#!/bin/bash
# round 3: a second long differential fuzz of the final kernels (other seed), then the text reader's
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03y; mkdir -p $O
cd $R
timeout 4200 python tools/fuzz_parity.py 160000 303910 > $O/fuzz_303910.txt 2>&1; echo "fuzz rc=$?"; tail -15 $O/fuzz_303910.txt
timeout 1500 python tools/text_fuzz.py 12000 20261008 > $O/text_fuzz.txt 2>&1; echo "text fuzz rc=$?"; tail -3 $O/text_fuzz.txt
