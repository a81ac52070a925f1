#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/smoke
rm -rf $O && mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
MOBOCMF_POISON=1 timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/poison_suite.log 2>&1
tail -3 $O/poison_suite.log
