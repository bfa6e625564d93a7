#!/bin/bash
# The GPU suite with one fast path switched off for EVERY test, knob by knob: the general kernels alone must give the
# same results (the tests that assert that a fast path RAN are the expected failures).  usage (GPU box):
# bash tools/knob_suite.sh OUTFILE
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/${1:-gpurun_out/knob_suite.txt}
mkdir -p $(dirname $OUT)
: > $OUT
cd $R
for kv in SBWT_NO_SETTLE=1 SBWT_NO_DIRECT=1 SBWT_NO_SHARED_LISTS=1 SBWT_NO_BFS=1 SBWT_NO_CHAIN=1 SBWT_NO_TWINS=1 SBWT_NO_SEGMENTS=1 \
          SBWT_NO_ROUNDS=1 SBWT_NO_RANK_PASS=1 SBWT_NO_SPECULATIVE_PENDING=1 SBWT_HASH_TEXT=0; do
  echo "== $kv" >> $OUT
  echo "== $kv"
  env $kv timeout -k 10 600 python -u -m pytest tests -m gpu -q --timeout 300 --timeout-method thread 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed|error" >> $OUT
  rc=${PIPESTATUS[0]}
  tail -1 $OUT
  # a time limit, an abort or a fault: nothing more on the GPU in this call
  if [ $rc -ge 124 ]; then echo "stopped: exit status $rc under $kv" | tee -a $OUT; exit 1; fi
done
