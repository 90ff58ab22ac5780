#!/bin/bash
# the whole GPU suite the driver runs at round end, with a log that names a hanging test
out=$PWD/gpurun_out/tests; mkdir -p $out
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $out/pytest_gpu.log 2>&1
rc=$?; tail -25 $out/pytest_gpu.log; exit $rc
