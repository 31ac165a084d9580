#!/bin/bash
# PMC counters of the memoising Merkle stage (stwo_merkle + stwo_top) on the default workload
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/d; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
B="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --inflight 1 --distinct 0 --e2e 0"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $O/pmc_valu -- $B > $O/pmc_valu.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_wait -- $B > $O/pmc_wait.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $B > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $B > $O/pmc_write.log 2>&1
grep -c . $O/pmc_*/*/*counter_collection.csv
