#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q "$@" > gpurun_out/pytest_gpu.log 2>&1; rc=$?
tail -30 gpurun_out/pytest_gpu.log
exit $rc
