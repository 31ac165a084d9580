#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/probe; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- $R/build/fetch_probe > $O/fetch.log 2>&1
grep -h "strided_rows\|scattered" $O/fetch/*/*_counter_collection.csv | awk -F'","' '{print $9, $16, $17}' | cut -c1-140
tail -1 $O/fetch.log
