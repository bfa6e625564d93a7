#!/bin/bash
# PMC passes over k_search at bench scale (one rocprofv3 run per pass, counters only).
# usage (GPU box): bash tools/pmc_k_search.sh OUTDIR [bench args...]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/${1:-gpurun_out/pmc}; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
n=0
for pass in "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
            "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
            "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU GRBM_GUI_ACTIVE GRBM_UTCL2_BUSY" \
            "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
            "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum" \
            "FETCH_SIZE" "WRITE_SIZE"; do
  n=$((n+1))
  timeout -s KILL 300 rocprofv3 --pmc $pass --kernel-include-regex 'k_search' --output-format csv -d $OUT/p$n -o bench -- python3 $R/bench.py --steps 2 --warmup 0 --no-cpu "$@" > $OUT/p$n.json 2> $OUT/p$n.err || { echo "pass $n failed"; tail -3 $OUT/p$n.err; exit 2; }
done
python3 - $OUT <<'PY'
import csv, collections, glob, sys
d = collections.defaultdict(list)
for f in sorted(glob.glob(sys.argv[1] + "/p*/bench_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        d[r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(sys.argv[1] + "/summary.txt", "w") as o:
    for k, v in d.items():
        line = "%-40s launches %d  mean %.6g" % (k, len(v), sum(v) / len(v))
        print(line); o.write(line + "\n")
PY
