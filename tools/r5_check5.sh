#!/bin/bash
set -u
mkdir -p gpurun_out
python tools/probe/probe_score_sigma.py 8 > gpurun_out/r5_score_sigma_cfg4s.json 2> gpurun_out/r5_score_sigma.err; echo "sigma rc=$?"; cat gpurun_out/r5_score_sigma_cfg4s.json
python -m pytest tests -x -q -m gpu > gpurun_out/r5e_pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$?"; tail -5 gpurun_out/r5e_pytest_gpu.log
