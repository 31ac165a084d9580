#!/bin/bash
# round 4: staging threads of the records paths, distinct source buffers
mkdir -p gpurun_out/r04f
for t in 4 8 12 16; do
  echo "== SS_STAGE_THREADS=$t" >> gpurun_out/r04f/stage_threads.txt
  SS_STAGE_THREADS=$t python tools/host_path_bench.py 8192 distinct 2>/dev/null | grep -v "^host path: 8192 proofs in 0.[1-9]" >> gpurun_out/r04f/stage_threads.txt
done
cat gpurun_out/r04f/stage_threads.txt
