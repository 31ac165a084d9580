#!/bin/bash
# Round-2 GPU pass B: pair memoisation -- parity suite, A/B benches, kernel stats.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/b; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
for w in stwo_2p20 stwo_2p16 stwo_wide256 stwo_fixture; do
  python bench.py --workload $w --steps 40 --warmup 4 --cpu-seconds 4 > $O/bench_$w.json 2> $O/bench_$w.err; echo "bench $w rc=$?"
  python bench.py --workload $w --steps 40 --warmup 4 --no-cpu-baseline --no-dedup > $O/bench_${w}_nodedup.json 2>> $O/bench_$w.err; echo "bench $w nodedup rc=$?"
done
python bench.py --workload stwo_2p20 --proofs-per-gpu 8192 --steps 100 --warmup 6 --no-cpu-baseline > $O/bench_stwo_2p20_8192.json 2>> $O/bench_stwo_2p20.err
build/sha_bench 512 > $O/sha_bench.txt 2>&1
cd /tmp; export TMPDIR=/tmp
B="python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B > $O/stats.log 2>&1
grep -h "stwo_\|Name" $O/stats/*/*_kernel_stats.csv | cut -c1-160
