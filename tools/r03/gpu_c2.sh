#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03c; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $O/prof -o e2e -- python3 $R/tools/e2e_bench.py --n 1536 --reps 2 --fmt json > $O/prof.log 2>&1
ls $O/prof | head -20
python3 - <<PY
import csv,glob
for f in sorted(glob.glob("$O/prof/*stats*.csv")):
    print(f)
    for i,r in enumerate(csv.reader(open(f))):
        if i<16: print(r)
PY
