#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r3t; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -6 $O/pytest.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
