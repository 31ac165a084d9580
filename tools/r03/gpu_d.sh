#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03d; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_text.py -q -x > $O/pytest_text.log 2>&1; echo "pytest rc=$?" >> $O/pytest_text.log
tail -5 $O/pytest_text.log
python tools/e2e_bench.py --n 1536 > $O/e2e_1536.json 2>$O/e2e.err; cat $O/e2e_1536.json
python tools/e2e_bench.py --n 8192 --reps 3 > $O/e2e_8192.json 2>>$O/e2e.err; cat $O/e2e_8192.json
python tools/e2e_bench.py --n 1536 --noncanonical 0.02 > $O/e2e_1536_nc.json 2>>$O/e2e.err; cat $O/e2e_1536_nc.json
