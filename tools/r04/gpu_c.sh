#!/bin/bash
# round 4: rates of the shared forms (text and records) beside the per-query forms
set -o pipefail
mkdir -p gpurun_out/r04d
python tools/e2e_bench.py --n 4096 --reps 5 --fmt all > gpurun_out/r04d/e2e_4096.json 2> gpurun_out/r04d/e2e_4096.err; tail -c 1800 gpurun_out/r04d/e2e_4096.json
python tools/e2e_bench.py --n 512 --reps 5 --fmt all > gpurun_out/r04d/e2e_512.json 2> gpurun_out/r04d/e2e_512.err
python tools/e2e_bench.py --n 4096 --reps 3 --fmt shared --files > gpurun_out/r04d/e2e_files.json 2> gpurun_out/r04d/e2e_files.err
python tools/host_path_bench.py 2048 > gpurun_out/r04d/host_path_2048.txt 2>&1
python tools/host_path_bench.py 16384 > gpurun_out/r04d/host_path_16384.txt 2>&1; tail -3 gpurun_out/r04d/host_path_16384.txt
