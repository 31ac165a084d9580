#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/i; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/prover_bench.py 20 4 > $O/prover_stats.log 2>&1
cut -c1-110 $O/stats/*/*_kernel_stats.csv | head -30
