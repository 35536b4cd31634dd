#!/bin/bash
O=gpurun_out/r3k; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_c5.py -x -q -m gpu -k "other_widths" > $O/tests.log 2>&1; grep -E "passed|failed|error" $O/tests.log | tail -3; grep -E "^FAILED|^E  " $O/tests.log | head -20
