#!/bin/bash
# round 4: where the prover's time goes (kernel stats of prove_many)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04h; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/prover_bench.py 20 2 sha256 4 32 > $O/prover.log 2>&1
tail -5 $O/prover.log
f=$(ls -t $O/stats/*/*_kernel_stats.csv | head -1); cp $f $O/kernel_stats.csv; head -30 $f | cut -c1-160
