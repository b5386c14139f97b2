#!/bin/bash
# Quick loop for work on the streaming front kernel: the GPU tests that exercise it, then its timing alone.
set -o pipefail
out=gpurun_out/${1:-front}
mkdir -p $out
timeout -k 10 600 python -u -m pytest tests/test_gpu_pipeline.py tests/test_gpu_gate.py tests/test_gpu_f1.py tests/test_gpu_ops.py -x -q -m gpu --timeout 300 --timeout-method thread > $out/gpu.log 2>&1
rc=$?
tail -3 $out/gpu.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python3 tools/front_probe.py 2>&1 | tail -3
