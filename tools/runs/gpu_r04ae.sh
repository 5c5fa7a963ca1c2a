#!/bin/bash
# round 4: full GPU suite + smoke on the final build
set -o pipefail
mkdir -p gpurun_out/r04ae
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r04ae/pytest.log 2>&1; rc=$?
tail -4 gpurun_out/r04ae/pytest.log
[ $rc -eq 0 ] || { tail -60 gpurun_out/r04ae/pytest.log; exit $rc; }
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
