#!/bin/bash
# round 4: shared-path texts through the GPU reader (tests)
set -o pipefail
mkdir -p gpurun_out/r04c
timeout -k 10 900 python -m pytest tests/test_gpu_shared.py tests/test_gpu_text.py -m gpu -x -q > gpurun_out/r04c/tests.log 2>&1
echo "rc=$?" >> gpurun_out/r04c/tests.log
tail -40 gpurun_out/r04c/tests.log
