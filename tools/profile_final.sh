#!/bin/bash
# Profile of the headline bench: rocprofv3 kernel trace + stats of the default command, then PMC passes (one
# rocprofv3 run each, counters only) restricted to the search kernel.
# usage (GPU box): bash tools/profile_final.sh OUTDIR
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/${1:-gpurun_out/prof_final}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 $R/bench.py > $OUT/bench_trace.json 2> $OUT/bench_trace.err || exit 1
n=0
for pass in "FETCH_SIZE" "WRITE_SIZE" \
            "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
            "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
            "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU GRBM_GUI_ACTIVE GRBM_UTCL2_BUSY" \
            "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum"; do
  n=$((n+1))
  timeout -s KILL 300 rocprofv3 --pmc $pass --kernel-include-regex 'k_search' --output-format csv -d $OUT/p$n -o bench -- python3 $R/bench.py --steps 2 --warmup 0 --no-cpu > $OUT/p$n.json 2> $OUT/p$n.err || { echo "pass $n failed"; tail -3 $OUT/p$n.err; exit 2; }
done
python3 - $OUT <<'PY'
import csv, collections, glob, sys
d = collections.defaultdict(list)
for f in sorted(glob.glob(sys.argv[1] + "/p*/bench_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        d[r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(sys.argv[1] + "/pmc_summary.txt", "w") as o:
    o.write("# per launch of k_search_sm (10^8 reads x 50 bp, GRCh38-sized index); mean over launches\n")
    for k, v in d.items():
        line = "%-40s launches %d  mean %.6g" % (k, len(v), sum(v) / len(v))
        print(line); o.write(line + "\n")
PY
grep k_search $OUT/trace/bench_kernel_stats.csv
