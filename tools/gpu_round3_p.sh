#!/bin/bash
timeout 1200 python -m pytest tests/test_gpu_bench_contract.py tests/test_gpu_config_scale.py -m gpu -x -q 2>&1 | tail -15
