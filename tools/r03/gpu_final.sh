#!/bin/bash
# Round-3 evidence pass on the final kernels: default bench line, every config's bench line, kernel stats, PMC counters,
# SHA calibration, host paths (records and text), prover.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03final; mkdir -p $O
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?"
for w in stwo_2p16 stwo_2p16_blake2s stwo_wide256 stwo_wide256_blake2s stwo_2p20_blake2s stwo_fixture; do
  python bench.py --workload $w --steps 60 --warmup 6 --cpu-seconds 4 --e2e 1024 > $O/bench_$w.json 2> $O/bench_$w.err; echo "bench $w rc=$?"
done
python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-dedup --e2e 0 > $O/bench_stwo_2p20_nodedup.json 2> $O/bench_nodedup.err
python bench.py --proofs-per-gpu 8192 --steps 200 --warmup 6 --no-cpu-baseline --e2e 0 > $O/bench_stwo_2p20_8192.json 2> $O/bench_8192.err
python bench.py --proofs-per-gpu 8192 --steps 200 --warmup 6 --no-cpu-baseline --e2e 0 --tail-streams 1 > $O/bench_stwo_2p20_8192_ts1.json 2>> $O/bench_8192.err
python bench.py --workload stark101 --steps 1920 --warmup 6 --cpu-seconds 3 > $O/bench_stark101_4096.json 2> $O/bench_stark101.err; echo "stark101 rc=$?"
python bench.py --workload stark101 --proofs-per-gpu 8192 --steps 960 --warmup 6 --no-cpu-baseline --e2e 0 > $O/bench_stark101_8192.json 2>> $O/bench_stark101.err
mkdir -p build
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/sha_bench.hip -o build/sha_bench 2>/dev/null && build/sha_bench 512 > $O/sha_bench.txt 2>&1
python tools/host_path_bench.py 2048 > $O/host_path.txt 2>&1; python tools/host_path_bench.py 16384 >> $O/host_path.txt 2>&1
python tools/e2e_bench.py --n 4096 --reps 4 --files > $O/e2e_4096.json 2> $O/e2e.err; echo "e2e rc=$?"
python tools/e2e_bench.py --n 4096 --reps 3 --noncanonical 0.01 > $O/e2e_4096_nc1.json 2>> $O/e2e.err
python tools/e2e_bench.py --n 512 --reps 4 > $O/e2e_512.json 2>> $O/e2e.err
python tools/e2e_bench.py --n 4096 --reps 3 --fmt json --python-separators > $O/e2e_4096_pysep.json 2>> $O/e2e.err
python tools/e2e_bench.py --n 4096 --reps 3 --workload stwo_trace16.npz > $O/e2e_4096_2p16.json 2>> $O/e2e.err
python tools/prover_bench.py 20 2 sha256 1,4,4,3 48 > $O/prover_bench.txt 2>&1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline --e2e 0 > $O/stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_e2e -- python3 $R/tools/e2e_bench.py --n 4096 --reps 2 --fmt json > $O/stats_e2e.log 2>&1
B="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --inflight 1 --distinct 0 --e2e 0 --tail-streams 1"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $O/pmc_valu -- $B > $O/pmc_valu.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_wait -- $B > $O/pmc_wait.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $B > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $B > $O/pmc_write.log 2>&1
E="python3 $R/tools/e2e_bench.py --n 1024 --reps 1 --fmt json"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_e2e -- $E > $O/pmc_fetch_e2e.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_e2e -- $E > $O/pmc_write_e2e.log 2>&1
cd $R
python - <<'PY'
import json,glob,os
O='gpurun_out/r03final'
for f in sorted(glob.glob(O+'/bench_*.json')):
    try: d=json.load(open(f))
    except Exception as e: print(f,'BAD',e); continue
    print(os.path.basename(f)[6:-5], round(d['value']), round(d['ms_per_step'],3), 'roof', round(d['roofline']['frac'],4), 'alu', round(d['alu_roofline']['frac'],4), {k:round(v,3) for k,v in d['kernels_ms_per_step'].items() if k in ('stwo_merkle','stwo_top','s101_merkle')}, 'e2e', {k:round(v['proofs_per_s']) for k,v in d.get('e2e',{}).items() if isinstance(v,dict)})
for f in sorted(glob.glob(O+'/e2e_*.json')):
    try: d=json.load(open(f))
    except Exception as e: print(f,'BAD',e); continue
    print(os.path.basename(f), {k:(round(v['proofs_per_s_best']),round(v['text_GB_per_s_best'],1),v.get('host_parsed'),round(v.get('files_proofs_per_s_best',0))) for k,v in d.items() if isinstance(v,dict)})
PY
tail -3 $O/host_path.txt; grep prove_many $O/prover_bench.txt
