#!/bin/bash
# one SQ pass over k_search at bench scale.  usage: bash tools/pmc_quick.sh OUTDIR [bench args]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/${1:-gpurun_out/pmcq}; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-include-regex 'k_search' --output-format csv -d $OUT/p -o bench -- python3 $R/bench.py --steps 2 --warmup 0 --no-cpu "$@" > $OUT/p.json 2> $OUT/p.err || { tail -3 $OUT/p.err; exit 2; }
python3 - $OUT <<'PY'
import csv, collections, glob, sys
d = collections.defaultdict(list)
for f in sorted(glob.glob(sys.argv[1] + "/p/bench_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        d[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in d.items():
    print("%-28s launches %d  mean %.6g" % (k, len(v), sum(v) / len(v)))
PY
tail -1 $OUT/p.json | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('k_search_ms', d['detail']['k_search_ms'])"
