#!/bin/bash
# Evidence pass of a round on the GPU box: bench lines, kernel statistics, PMC counters, calibration, host paths,
# prover, differential fuzzers.  One parametrised script (rounds 2 and 3 kept a script per experiment):
#
#   gpurun --timeout 1200 -- 'bash tools/evidence.sh r04 [section ...]'      then here:  python tools/profiles.py r04
#
# sections (default: all): bench configs stats pmc e2e host sha prover fuzz queues
# Output: gpurun_out/<tag>/ (scratch); tools/profiles.py turns it into the committed profiles/<tag>_* files.
TAG=${1:?usage: evidence.sh <tag> [sections]}; shift
SECTIONS=${*:-bench configs stats pmc e2e host sha prover fuzz queues}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
want() { [[ " $SECTIONS " == *" $1 "* ]]; }
FAILED=0
ok() { local rc=$?; echo "$1 rc=$rc"; [ $rc -eq 0 ] || FAILED=$((FAILED + 1)); }
# the runtime reads this when it initialises -- under rocprofv3 that is before bench.py's own setdefault runs, so the profiler
# passes would otherwise see 4 hardware queues and another stream-to-queue mapping than the benchmark
export GPU_MAX_HW_QUEUES=24
# what was measured: the digest of the kernel sources of THIS tree (the GPU box's copy), for tools/profiles.py to stamp the
# profiles with -- not the digest of whatever tree the summary is later made in
python3 -c "import bench, json; json.dump({'kernel_sources_sha256': bench.kernel_sources_digest()}, open('$O/measured_tree.json', 'w'))"
B="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --inflight 1 --distinct 0 --e2e 0 --tail-streams 1"

if want bench; then
  python bench.py > $O/bench_default.json 2> $O/bench_default.err; ok default
fi
if want configs; then
  for w in stwo_2p16 stwo_2p16_blake2s stwo_wide256 stwo_wide256_blake2s stwo_2p20_blake2s stwo_fixture; do
    python bench.py --workload $w --steps 60 --warmup 6 --cpu-seconds 4 --e2e 1024 > $O/bench_$w.json 2> $O/bench_$w.err; ok "bench $w"
  done
  python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-dedup --e2e 0 > $O/bench_stwo_2p20_nodedup.json 2> $O/bench_nodedup.err
  python bench.py --proofs-per-gpu 8192 --steps 200 --warmup 6 --no-cpu-baseline --e2e 0 > $O/bench_stwo_2p20_8192.json 2> $O/bench_8192.err
  python bench.py --proofs-per-gpu 8192 --steps 200 --warmup 6 --no-cpu-baseline --e2e 0 --tail-streams 1 --inflight 3 > $O/bench_stwo_2p20_8192_ts1.json 2>> $O/bench_8192.err
  python bench.py --steps 40 --warmup 4 --no-cpu-baseline --e2e 0 --tail-streams 1 > $O/bench_stwo_2p20_ts1.json 2>> $O/bench_8192.err
  python bench.py --workload stark101 --steps 1920 --warmup 6 --cpu-seconds 3 > $O/bench_stark101_4096.json 2> $O/bench_stark101.err; ok stark101
  python bench.py --workload stark101 --proofs-per-gpu 8192 --steps 960 --warmup 6 --no-cpu-baseline --e2e 0 > $O/bench_stark101_8192.json 2>> $O/bench_stark101.err
fi
if want sha; then
  mkdir -p build
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/sha_bench.hip -o build/sha_bench 2>/dev/null && build/sha_bench 512 > $O/sha_bench.txt 2>&1
fi
if want host; then
  rm -f $O/host_path.txt
  for n in 2048 16384; do
    python tools/host_path_bench.py $n >> $O/host_path.txt 2>&1
    python tools/host_path_bench.py $n distinct >> $O/host_path.txt 2>&1
  done
fi
if want queues; then  # ss_process_defaults() from a C++ program that links the library (csrc/ss_env.cpp)
  mkdir -p build
  /opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 -Iinclude tools/probes/hw_queues_probe.hip -o build/hw_queues_probe -Lstark-symphony_amd -lss_verify -Wl,-rpath,$R/stark-symphony_amd 2>/dev/null
  P="build/hw_queues_probe tests/golden/stark101_proof.json"
  ( unset GPU_MAX_HW_QUEUES
    for i in 1 2; do
      echo -n "ss_process_defaults() called:     "; $P 2>&1 | grep -v amdgpu.ids
      echo -n "not called (runtime default):     "; $P nodefaults 2>&1 | grep -v amdgpu.ids
      echo -n "SS_KEEP_ENV=1 (runtime default):  "; SS_KEEP_ENV=1 $P 2>&1 | grep -v amdgpu.ids
      echo -n "caller sets GPU_MAX_HW_QUEUES=24: "; GPU_MAX_HW_QUEUES=24 $P 2>&1 | grep -v amdgpu.ids
    done ) > $O/hw_queues.txt 2>&1; ok queues
fi
if want e2e; then
  python tools/e2e_bench.py --n 4096 --reps 4 --fmt all --files > $O/e2e_4096.json 2> $O/e2e.err; ok e2e
  python tools/e2e_bench.py --n 4096 --reps 3 --fmt all --noncanonical 0.01 > $O/e2e_4096_nc1.json 2>> $O/e2e.err
  python tools/e2e_bench.py --n 512 --reps 4 --fmt all > $O/e2e_512.json 2>> $O/e2e.err
  python tools/e2e_bench.py --n 4096 --reps 3 --fmt json --python-separators > $O/e2e_4096_pysep.json 2>> $O/e2e.err
  python tools/e2e_bench.py --n 4096 --reps 3 --fmt all --workload stwo_trace16.npz > $O/e2e_4096_2p16.json 2>> $O/e2e.err
fi
if want prover; then
  python tools/prover_bench.py 20 3 sha256 1,4,4,3 48 > $O/prover_bench.txt 2>&1
  python tools/prover101_bench.py >> $O/prover_bench.txt 2>&1
fi
if want fuzz; then
  python tools/fuzz_parity.py 20000 20261004 > $O/fuzz_parity.txt 2>&1; ok fuzz
  python tools/text_fuzz.py 4000 20261004 > $O/text_fuzz.txt 2>&1; ok "text fuzz"
  python tools/shape_sweep.py 100 4 > $O/shape_sweep.txt 2>&1; ok "shape sweep"
fi
# the profiler passes last, from /tmp (rocprofv3 writes beside its working directory); the program itself after `--`
cd /tmp; export TMPDIR=/tmp
if want stats; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline --e2e 0 > $O/stats.log 2>&1; ok stats
  # the default submission overlaps consecutive launches of the Merkle stage (two tail streams): a kernel's duration in that trace is
  # not its own.  The roofline's durations come from a pass through a ONE-slot pipeline (nothing overlaps) -- this is its rocprof twin.
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_ts1 -- python3 $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline --e2e 0 --tail-streams 1 --inflight 1 > $O/stats_ts1.log 2>&1; ok stats_ts1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_e2e -- python3 $R/tools/e2e_bench.py --n 4096 --reps 2 --fmt all > $O/stats_e2e.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_prover -- python3 $R/tools/prover_bench.py 20 3 sha256 > $O/stats_prover.log 2>&1
fi
if want pmc; then  # counters in their own runs, one set per pass (MI355X_MICROARCH.md)
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $O/pmc_valu -- $B > $O/pmc_valu.log 2>&1; ok pmc_valu
  rocprofv3 --pmc SQ_WAVES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_wait -- $B > $O/pmc_wait.log 2>&1; ok pmc_wait
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $B > $O/pmc_fetch.log 2>&1; ok pmc_fetch
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $B > $O/pmc_write.log 2>&1; ok pmc_write
  E="python3 $R/tools/e2e_bench.py --n 1024 --reps 1 --fmt all"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_e2e -- $E > $O/pmc_fetch_e2e.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_e2e -- $E > $O/pmc_write_e2e.log 2>&1
fi
cd $R
python - "$O" <<'PY'
import glob, json, os, sys
O = sys.argv[1]
for f in sorted(glob.glob(O + '/bench_*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
    except Exception as e:
        print(f, 'BAD', e)
        continue
    print(os.path.basename(f)[6:-5], round(d['value']), round(d['ms_per_step'], 3), 'roof', round(d['roofline']['frac'], 4), 'alu',
          round(d['alu_roofline']['frac'], 4), 'issue', d['alu_roofline'].get('issue_frac'),
          {k: round(v, 3) for k, v in d['kernels_ms_per_step'].items() if k in ('stwo_merkle', 'stwo_top', 's101_merkle')},
          'e2e', {k: round(v['proofs_per_s']) for k, v in d.get('e2e', {}).items() if isinstance(v, dict) and 'proofs_per_s' in v})
for f in sorted(glob.glob(O + '/e2e_*.json')):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, 'BAD', e)
        continue
    print(os.path.basename(f), {k: (round(v['proofs_per_s_best']), round(v['text_GB_per_s_best'], 1), v.get('host_parsed'),
                                    round(v.get('files_proofs_per_s_best') or 0)) for k, v in d.items() if isinstance(v, dict)})
PY
[ -f $O/host_path.txt ] && grep -v amdgpu.ids $O/host_path.txt | tail -8
[ -f $O/prover_bench.txt ] && grep prove_many $O/prover_bench.txt
[ -f $O/fuzz_parity.txt ] && tail -1 $O/fuzz_parity.txt
[ -f $O/text_fuzz.txt ] && tail -1 $O/text_fuzz.txt
[ -f $O/shape_sweep.txt ] && tail -1 $O/shape_sweep.txt
echo "sections that failed: $FAILED"
[ $FAILED -eq 0 ]
