#!/bin/bash
# GPU_MAX_HW_QUEUES policy with data: the three submissions that depend on it, at 4 / 8 / 12 / 16 / 24 / 32
# hardware queues, plus what a foreign stream of the process pays for the queues in use.
#   gpurun --timeout 1100 -- 'bash tools/hw_queues_sweep.sh > gpurun_out/r06/hw_queues_sweep.txt 2>&1'
cd ${GRAFT_REPO_ROOT:-$(pwd)}
line() { python3 -c "
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print('%-34s q=%-7s %12.0f proofs/s  %8.3f ms/step  alu %.3f  %s' % (sys.argv[2], sys.argv[3], d['value'], d['ms_per_step'], d['alu_roofline']['frac'], d['config']['submission'][:60]))" "$@"; }
T=$(mktemp -d)
for q in 4 8 12 16 24 32; do
  export GPU_MAX_HW_QUEUES=$q
  python bench.py --steps 40 --warmup 4 --no-cpu-baseline --e2e 0 --distinct 16 > $T/a.json 2> $T/a.err && line $T/a.json "stwo 2^20 x 65536 (metric)" $q || tail -3 $T/a.err
  SS_BENCH_GROUP_OF_ONE=1 python bench.py --proofs-per-gpu 8192 --steps 200 --warmup 6 --no-cpu-baseline --e2e 0 --distinct 16 > $T/b.json 2> $T/b.err && line $T/b.json "stwo 2^20 x 8192 + RCCL reduce" $q || tail -3 $T/b.err
  python bench.py --workload stark101 --steps 1920 --warmup 6 --no-cpu-baseline --e2e 0 > $T/c.json 2> $T/c.err && line $T/c.json "stark101 x 4096, 16 streams" $q || tail -3 $T/c.err
  python tools/probes/cross_queue_latency.py 2>&1 | grep -v amdgpu.ids
done
