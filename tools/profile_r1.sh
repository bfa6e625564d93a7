#!/bin/bash
# Round-1 profile of the headline bench: kernel trace + stats, then PMC passes restricted to k_search.
# usage (GPU box): bash tools/profile_r1.sh OUTDIR
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/${1:-gpurun_out/prof_r1}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 $R/bench.py --steps 3 --warmup 1 > $OUT/bench_trace.json 2> $OUT/bench_trace.err || exit 1
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  tag=$(echo $pass | tr ' ' '_')
  timeout -s KILL 300 rocprofv3 --pmc $pass --kernel-include-regex 'k_search' --output-format csv -d $OUT/pmc_$tag -o bench -- python3 $R/bench.py --steps 2 --warmup 0 --no-cpu > $OUT/pmc_$tag.json 2> $OUT/pmc_$tag.err || exit 2
done
