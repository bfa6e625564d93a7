#!/bin/bash
# The bench lines of DESIGN.md section 4 beside the headline: tools/side_benches.sh OUTDIR   (GPU box; one JSON line per file)
# Stops at the first run that fails.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/$1
mkdir -p $OUT
run() {
  name=$1; shift
  timeout -k 10 500 python3 $R/bench.py --no-extras "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err || { echo "$name failed"; tail -3 $OUT/bench_$name.err; exit 1; }
  python3 - $OUT/bench_$name.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("%-28s %.3e reads/s  step %.3f ms  %s %.3f ms  lines/read %.3f  parity %s" % (
    sys.argv[1].split("bench_")[-1][:-5], d["value"], d["ms_per_step"], r.get("kernel"), r.get("launch_ms") or 0,
    r["whole_step"]["index_lines_per_read"], (d["detail"].get("parity_vs_oracle") or {}).get("identical")))
PY
}
run L50_compact --layout compact &&
run L50_plain --layout plain &&
run L100_hits --read-len 100 &&
run L100_compact --read-len 100 --layout compact &&
run L150 --read-len 150 --reads 50000000 &&
run L150_compact --read-len 150 --reads 50000000 --layout compact &&
run mutate20 --mutate 0.2 &&
run mutate20_L100 --mutate 0.2 --read-len 100 &&
run dm6 --scale 0.0486 --reads 10000000 --read-len 36 &&
run ecoli --scale 0.00158 --reads 10000000 &&
run smallrna --reads 20000000 --read-len 30 --min-len 18 --repeats &&
run rank_path --rank-path --layout compact &&
run repeats --genome repeats --reads 20000000 &&
run repeats_shuffled --genome repeats --reads 20000000 --shuffle-ties
