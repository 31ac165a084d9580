#!/bin/bash
# round 3, run J: where the 256-column shape loses its 17 %: per-kernel counters, no overlap between kernels
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03j; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
B="python3 $R/bench.py --workload stwo_wide256 --steps 3 --warmup 1 --no-cpu-baseline --inflight 1 --distinct 0 --e2e 0 --tail-streams 1"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $O/pmc_valu -- $B > $O/pmc_valu.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_wait -- $B > $O/pmc_wait.log 2>&1
cd $R
python tools/pmc_summary.py $O/pmc_valu $O/pmc_wait > $O/pmc_wide256.json
python - <<PY
import json
d=json.load(open('$O/pmc_wide256.json'))
for k,v in d.items():
    if not k.startswith('stwo_'): continue
    g=v.get('GRBM_GUI_ACTIVE',{}).get('avg',0)/8
    iv=v.get('SQ_INSTS_VALU',{}).get('avg',0)
    wc=v.get('SQ_WAVE_CYCLES',{}).get('avg',0)
    print("%-28s ms %.3f  VALU %.3f G  gpu_cycles %.2f M  util %.3f  waves/SIMD %.2f  wait_any/wave_cycles %.2f"%(k, v['avg_ms_with_counters'], iv/1e9, g/1e6, iv/(1024*g/4) if g else 0, wc*4/(1024*g) if g else 0, v.get('SQ_WAIT_ANY',{}).get('avg',0)/max(wc,1)))
PY
