#!/bin/bash
# round 3, run C: where the text -> verdict time goes
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03c; mkdir -p $O
cd $R
python tools/e2e_bench.py --n 1536 > $O/e2e_1536.json 2>$O/e2e.err; cat $O/e2e_1536.json
python tools/e2e_bench.py --n 8192 --reps 3 > $O/e2e_8192.json 2>>$O/e2e.err; cat $O/e2e_8192.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --stats -d $O/prof -o e2e -- python3 $R/tools/e2e_bench.py --n 1536 --reps 2 --fmt json > $O/prof.log 2>&1
ls $O/prof | head; 
python3 - <<PY
import csv,glob
for f in glob.glob("$O/prof/*kernel_stats.csv")+glob.glob("$O/prof/*memory_copy_stats.csv"):
    print(f)
    for i,r in enumerate(csv.reader(open(f))):
        if i<14: print(r)
PY
