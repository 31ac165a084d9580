#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/e; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "memoisation or baseline_configs or wrong_path or fuzz" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
for w in stwo_2p20 stwo_2p16 stwo_wide256 stwo_fixture stwo_2p16_blake2s; do
  python bench.py --workload $w --steps 40 --warmup 4 --no-cpu-baseline --e2e 0 > $O/bench_$w.json 2> $O/bench_$w.err; echo "bench $w rc=$?"
done
for f in $O/bench_*.json; do python -c "
import json
d=json.load(open('$f')); k=d['kernels_ms_per_step']; print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],3),'ms merkle', round(k.get('stwo_merkle',0),2), 'top', round(k.get('stwo_top',0),2), round(d['alu_roofline']['frac'],3))"; done
cd /tmp; export TMPDIR=/tmp
B="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --inflight 1 --distinct 0 --e2e 0"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $B > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc_valu -- $B > $O/pmc_valu.log 2>&1
