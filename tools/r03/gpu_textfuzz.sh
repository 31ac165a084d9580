#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03textfuzz; mkdir -p $O
cd $R
timeout 2400 python tools/text_fuzz.py ${1:-20000} ${2:-20261007} > $O/text_fuzz.txt 2>&1; echo "text fuzz rc=$?"
grep -v amdgpu $O/text_fuzz.txt | tail -16
