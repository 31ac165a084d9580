#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03p; mkdir -p $O
cd $R
timeout 3000 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
