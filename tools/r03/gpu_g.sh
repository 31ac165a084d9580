#!/bin/bash
# round 3, run G: new top-sibling layout + cold kernel: parity, then top-kernel variants (waves/SIMD x light checks per hash)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03g; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_intermediates.py -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -6 $O/pytest.log
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function"
for v in "3 2" "4 2" "4 1" "3 1"; do
  set -- $v
  touch stark-symphony_amd/csrc/ss_stwo.hip
  make -C stark-symphony_amd/csrc HIPFLAGS="$BASE -DSS_TOP_WAVES=$1 -DSS_TOP_LIGHTS=$2" > $O/make_$1_$2.log 2>&1 || { echo "make failed $v"; tail -5 $O/make_$1_$2.log; continue; }
  python bench.py --steps 20 --warmup 4 --no-cpu-baseline --e2e 0 > $O/bench_$1_$2.json 2> $O/bench_$1_$2.err || { echo "bench failed $v"; tail -5 $O/bench_$1_$2.err; continue; }
  python - <<PY
import json
d=json.load(open('$O/bench_$1_$2.json')); print("waves=$1 lights=$2", round(d['value']), round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernels_ms_per_step'].items()}, round(d['alu_roofline']['frac'],4))
PY
done
