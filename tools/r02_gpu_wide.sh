#!/bin/bash
# instruction budget of the 256-column shape per kernel
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/wide; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "independent_streams or simfony_run_shim" 2>&1 | tail -3
cd /tmp; export TMPDIR=/tmp
B="python3 $R/bench.py --workload stwo_wide256 --steps 3 --warmup 1 --no-cpu-baseline --inflight 1 --distinct 0 --e2e 0"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES --kernel-trace --output-format csv -d $O/pmc_valu -- $B > $O/pmc_valu.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_wait -- $B > $O/pmc_wait.log 2>&1
