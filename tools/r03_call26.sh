#!/bin/bash
O=gpurun_out/r3aj; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/tests_full.log 2>&1; tail -5 $O/tests_full.log; grep -E "^FAILED|^ERROR" $O/tests_full.log | head
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
