#!/bin/bash
# round 4, first pass: shared records on the GPU (tests + host-path rates)
set -o pipefail
mkdir -p gpurun_out/r04b
python -m pytest tests/test_gpu_shared.py -m gpu -x -q > gpurun_out/r04b/shared_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r04b/shared_tests.log
tail -15 gpurun_out/r04b/shared_tests.log
python tools/host_path_bench.py 2048 > gpurun_out/r04b/host_path_2048.txt 2>&1; tail -6 gpurun_out/r04b/host_path_2048.txt
python tools/host_path_bench.py 16384 > gpurun_out/r04b/host_path_16384.txt 2>&1; tail -6 gpurun_out/r04b/host_path_16384.txt
