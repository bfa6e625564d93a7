#!/bin/bash
# Kernel trace + stats of one bench.py run: tools/trace_kernels.sh OUTDIR [bench.py arguments]
# (GPU box; writes OUTDIR/trace/*_kernel_stats.csv and OUTDIR/bench.json)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/$1
shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 $R/bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
find $OUT/trace -name "*kernel_trace.csv" -size +8M -delete   # gpurun_out travels back only below 64 MiB
f=$(ls $OUT/trace/*kernel_stats.csv | head -1)
cut -d, -f1-8 $f | head -14
