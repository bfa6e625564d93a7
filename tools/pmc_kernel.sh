#!/bin/bash
# PMC passes (one rocprofv3 run each, counters only) over the kernels matching a regex, at bench scale.
# usage (GPU box): [PMC_PASSES="1 2 5"] bash tools/pmc_kernel.sh OUTDIR KERNEL_REGEX [bench args...]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/$1; RX=$2; shift; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
n=0
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_LDS" \
            "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
            "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
            "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR" \
            "FETCH_SIZE" "WRITE_SIZE"; do
  n=$((n+1))
  if [ -n "${PMC_PASSES:-}" ] && ! [[ " $PMC_PASSES " == *" $n "* ]]; then continue; fi   # PMC_PASSES="1 2 4": only those passes
  timeout -s KILL 300 rocprofv3 --pmc $pass --kernel-include-regex "$RX" --output-format csv -d $OUT/p$n -o bench -- python3 $R/bench.py --steps 2 --warmup 0 --no-cpu --no-extras "$@" > $OUT/p$n.json 2> $OUT/p$n.err || { echo "pass $n ($pass) failed"; tail -3 $OUT/p$n.err; }
done
python3 - $OUT "$RX" "$*" <<'PY'
import csv, collections, glob, sys
d = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(sys.argv[1] + "/p*/bench_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(sys.argv[1] + "/pmc_summary.txt", "w") as o:
    o.write("# rocprofv3 --pmc, kernels matching %r, bench args %r; mean per launch\n" % (sys.argv[2], sys.argv[3]))
    for kn, cs in d.items():
        o.write("## %s\n" % kn)
        for k, v in cs.items():
            line = "%-40s launches %d  mean %.6g" % (k, len(v), sum(v) / len(v))
            o.write(line + "\n")
print(open(sys.argv[1] + "/pmc_summary.txt").read())
PY
