#!/bin/bash
# usage (on the GPU box): bash tools/ab_env.sh <outdir> "<bench args>" "ENV1=a ENV2=b" "ENV1=c" ...
# Runs bench.py once per environment setting (same box, back to back) and prints ms/step + the roofline fraction of each.
OUT=$1; shift
ARGS=$1; shift
mkdir -p $OUT
i=0
for E in "$@"; do
  tag=$(echo "$E" | tr ' =/' '___')
  env $E python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 $ARGS --detail $OUT/detail_$tag.json > $OUT/line_$tag.json 2> $OUT/err_$tag.txt
  python3 - "$OUT/line_$tag.json" "$E" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("%-40s %8.3f ms/step  %7.2f tiles/s  frac %.4f  avg_launch %.4f ms" % (sys.argv[2], d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"]), flush=True)
except Exception as e:
    print("%-40s FAILED %s" % (sys.argv[2], e), flush=True)
PY
  i=$((i+1))
done
