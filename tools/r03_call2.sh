#!/bin/bash
O=gpurun_out/r3b; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/tests_full.log 2>&1; tail -5 $O/tests_full.log
python tools/r03_margins.py > $O/margins.txt 2>&1; cat $O/margins.txt
