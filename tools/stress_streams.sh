#!/bin/bash
# Ordering stress of the multi-stream helpers: every test marked `streams` (pipeline, independent streams,
# graph replay, two contexts / two threads, the chunked host entry point) N times in ONE process, under the library's default
# of 24 hardware queues, under the runtime's 4, and with the library's default switched off.
#   gpurun --timeout 1200 -- 'bash tools/stress_streams.sh 50 > gpurun_out/stress.txt 2>&1'
N=${1:-50}
cd ${GRAFT_REPO_ROOT:-$(pwd)}
FAILED=0
run() {
  echo "=== $1 (repeat $N) ==="
  env $2 python -m pytest tests -m "gpu and streams" --repeat-streams $N -x -q -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -4
  [ ${PIPESTATUS[0]} -eq 0 ] || FAILED=$((FAILED + 1))
}
run "default (the binding asks for 24 hardware queues)" "SS_NOP=1"
run "GPU_MAX_HW_QUEUES=4" "GPU_MAX_HW_QUEUES=4"
run "SS_KEEP_ENV=1 (runtime default)" "SS_KEEP_ENV=1"
run "GPU_MAX_HW_QUEUES=32" "GPU_MAX_HW_QUEUES=32"
echo "configurations that failed: $FAILED"
[ $FAILED -eq 0 ]
