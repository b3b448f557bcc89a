#!/bin/bash
# round 4, call 37: smoke() and three repeats of the pairing / ops / check tests on the final library (flakiness check: gate, order tables)
cd "$(dirname "$0")/.."
O=gpurun_out/r4_37; mkdir -p $O
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -2 $O/smoke.log
for rep in 1 2 3; do
timeout 1200 python -m pytest tests/test_parity_gpu.py tests/test_ops_gpu.py -m gpu -x -q -k "pairing or ops" > $O/pytest_$rep.log 2>&1; echo "pytest $rep rc $?"; tail -2 $O/pytest_$rep.log
done
