#!/bin/bash
# round 3: one process per proof, as the reference's `make run` does: wall time of examples/ss_run for ONE witness
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03run1; mkdir -p $O
cd $R
gcc -O2 -Iinclude examples/ss_run.c -o $O/ss_run -Lstark-symphony_amd -lss_verify -Wl,-rpath,$R/stark-symphony_amd || exit 1
F=tests/golden/formats
for i in 1 2 3; do
  s=$(date +%s.%N); $O/ss_run run stark101/main.simf --witness $F/stark101_proof.wit > /dev/null 2>&1; rc=$?; e=$(date +%s.%N); echo "stark101 wit: rc $rc, $(echo "$e - $s" | bc) s wall"
done
for i in 1 2 3; do
  s=$(date +%s.%N); $O/ss_run run stwo-verifier/main.simf --witness $F/stwo_proof.wit > /dev/null 2>&1; rc=$?; e=$(date +%s.%N); echo "stwo wit: rc $rc, $(echo "$e - $s" | bc) s wall"
done
s=$(date +%s.%N); SS_TRACE_INIT=1 $O/ss_run run stwo-verifier/main.simf --witness $F/stwo_proof.wit; e=$(date +%s.%N); echo "stwo wit (output shown): $(echo "$e - $s" | bc) s wall"
python3 - <<PY
import time,subprocess
t=time.perf_counter(); subprocess.run(["$O/ss_run","run","stwo-verifier/main.simf","--witness","$F/stwo_proof.wit"],capture_output=True); print("python-timed: %.3f s" % (time.perf_counter()-t))
PY
