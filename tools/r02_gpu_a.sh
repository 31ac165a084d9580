#!/bin/bash
# Round-2 GPU pass A: full GPU suite, default bench, SHA calibration + LDS A/B, PMC counter passes.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/a; mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
build/sha_bench 512 > $O/sha_bench.txt 2>&1
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L > $O/counters_avail.txt 2>&1
B="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --inflight 1 --distinct 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B > $O/stats.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_valu -- $B > $O/pmc_valu.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_wait -- $B > $O/pmc_wait.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $B > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $B > $O/pmc_write.log 2>&1
find $O -name "*.csv" -size +20M -delete
du -sh $O
