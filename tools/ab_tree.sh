#!/bin/bash
# Same-box A/B of the working tree against a full copy of an earlier revision in abtest/prev (python + library):
#   here:  rm -rf abtest/prev && mkdir -p abtest/prev && git archive <rev> | tar -x -C abtest/prev && bash abtest/prev/mobocmf_amd/csrc/build.sh
#   box:   gpurun -- 'bash tools/ab_tree.sh "--config C5" "--surrogates 1" ""'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for args in "$@"; do
  for i in 1 2; do
    for tree in abtest/prev .; do
      v=$(cd $tree && timeout -k 10 300 python bench.py $args --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['value'], 1))") || exit 1
      echo "[$args] $tree $v"
    done
  done
done
