#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/stg2
timeout 1500 python -m pytest tests/test_focf_hip.py -x -q -m gpu > gpurun_out/stg2/pytest.log 2>&1
tail -5 gpurun_out/stg2/pytest.log
