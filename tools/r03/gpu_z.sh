#!/bin/bash
# round 3: per-kernel time of one 2^20-row proof after ss_p_lde / x-only composition columns
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03z; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/prover_bench.py 20 8 > $O/stats.log 2>&1
cd $R
python - <<'PY'
import csv,glob
f=sorted(glob.glob('gpurun_out/r03z/stats/*/*_kernel_stats.csv'))[-1]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:16]:
    print("%-70s calls %6s  total %8.2f ms  per proof %6.3f ms  %5.1f%%" % (r['Name'][:70], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['TotalDurationNs'])/1e6/9, 100*float(r['TotalDurationNs'])/tot))
print("sum per proof %.2f ms" % (tot/1e6/9))
PY
