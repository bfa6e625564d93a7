#!/bin/bash
# one PMC pass over k_search at bench scale.  usage: bash tools/pmc_pass.sh OUTDIR "COUNTERS..." [bench args]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/$1; CNT=$2; shift; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc $CNT --kernel-include-regex 'k_search' --output-format csv -d $OUT/p -o bench -- python3 $R/bench.py --steps 2 --warmup 0 --no-cpu "$@" > $OUT/p.json 2> $OUT/p.err || { tail -3 $OUT/p.err; exit 2; }
python3 - $OUT <<'PY'
import csv, collections, glob, sys
d = collections.defaultdict(list)
for f in sorted(glob.glob(sys.argv[1] + "/p/bench_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        d[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in d.items():
    print("%-40s launches %d  mean %.6g" % (k, len(v), sum(v) / len(v)))
PY
