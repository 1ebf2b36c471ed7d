#!/bin/bash
timeout 1200 python -m pytest tests/test_gpu_sharded.py -m gpu -x -q 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
