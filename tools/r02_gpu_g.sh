#!/bin/bash
# Round-2 evidence pass on the final kernels: kernel stats, PMC counters, every config's bench line,
# SHA calibration + LDS A/B, PCIe-inclusive host path, long fuzz run.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/g; mkdir -p $O
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?"
for w in stwo_2p16 stwo_2p16_blake2s stwo_wide256 stwo_wide256_blake2s stwo_2p20_blake2s stwo_fixture; do
  python bench.py --workload $w --steps 60 --warmup 6 --cpu-seconds 4 > $O/bench_$w.json 2> $O/bench_$w.err; echo "bench $w rc=$?"
done
for w in stwo_2p20 stwo_2p16 stwo_2p16_blake2s; do
  python bench.py --workload $w --steps 40 --warmup 4 --no-cpu-baseline --no-dedup --e2e 0 > $O/bench_${w}_nodedup.json 2>> $O/bench_$w.err
done
python bench.py --proofs-per-gpu 8192 --steps 100 --warmup 6 --no-cpu-baseline --e2e 0 > $O/bench_stwo_2p20_8192.json 2> $O/bench_8192.err
python bench.py --workload stark101 --steps 1920 --warmup 6 --cpu-seconds 3 > $O/bench_stark101_4096.json 2> $O/bench_stark101.err
python bench.py --workload stark101 --proofs-per-gpu 8192 --steps 960 --warmup 6 --no-cpu-baseline > $O/bench_stark101_8192.json 2>> $O/bench_stark101.err
build/sha_bench 512 > $O/sha_bench.txt 2>&1
python tools/host_path_bench.py 2048 > $O/host_path.txt 2>&1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline --e2e 0 > $O/stats.log 2>&1
B="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --inflight 1 --distinct 0 --e2e 0"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $O/pmc_valu -- $B > $O/pmc_valu.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_wait -- $B > $O/pmc_wait.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $B > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $B > $O/pmc_write.log 2>&1
cd $R
timeout 1500 python tools/fuzz_parity.py 8000 20261003 > $O/fuzz.txt 2>&1; echo "fuzz rc=$?"
tail -3 $O/fuzz.txt
