#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/poison; mkdir -p $O
MOBOCMF_POISON=1 timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/poison_suite.log 2>&1
tail -4 $O/poison_suite.log
