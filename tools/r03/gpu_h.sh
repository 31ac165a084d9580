#!/bin/bash
# round 3, run H: SHA-256 formulations; HBM traffic of the Merkle stage with the per-proof top-sibling layout
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03h; mkdir -p $O
cd $R
mkdir -p build
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/sha_formulations.hip -o build/sha_form 2> $O/sha_form_build.log && build/sha_form 256 > $O/sha_formulations.txt 2>&1
cat $O/sha_formulations.txt
cd /tmp; export TMPDIR=/tmp
B="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --inflight 1 --distinct 0 --e2e 0"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $B > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $B > $O/pmc_write.log 2>&1
cd $R
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write > $O/pmc_traffic.json; python - <<PY
import json
d=json.load(open('$O/pmc_traffic.json'))
tot=0
for k in ('stwo_merkle_kernel_sha','stwo_top_kernel_sha'):
    f=d[k]['FETCH_SIZE']['avg']*2048; w=d[k]['WRITE_SIZE']['avg']*1024
    print(k, 'fetch GB', f/1e9, 'write GB', w/1e9); tot+=f+w
print('stage', tot/1e9, 'ratio', tot/(170296*65536))
PY
