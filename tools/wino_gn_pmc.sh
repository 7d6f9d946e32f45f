#!/bin/bash
# usage (GPU box): bash tools/wino_gn_pmc.sh <outdir-under-gpurun_out> "<B H C K>" gn1 gn2 ...
# HBM-side traffic (FETCH_SIZE x2 + WRITE_SIZE, MI355X_MICROARCH.md) and duration of one Winograd launch shape per workgroup order
# (UNET_WINO_GN = n-tiles per group: consecutive slots walk gn n-tiles of one m-tile, then the next m-tile).
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
SHAPE=$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for GN in "$@"; do
  export UNET_WINO_GN=$GN
  tag=$(echo "$SHAPE" | tr ' ' '_')_gn$GN
  timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d $OUT/f_$tag --pmc FETCH_SIZE -- python3 $GRAFT_REPO_ROOT/tools/wino_one.py $SHAPE > $OUT/f_$tag.log 2>&1
  timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d $OUT/w_$tag --pmc WRITE_SIZE -- python3 $GRAFT_REPO_ROOT/tools/wino_one.py $SHAPE > $OUT/w_$tag.log 2>&1
  python3 - $OUT $tag "$SHAPE" $GN <<'PY'
import csv, glob, sys
out, tag, shape, gn = sys.argv[1:5]
def total(sub, counter):
    v = n = 0
    for f in glob.glob("%s/%s_%s/**/*counter_collection.csv" % (out, sub, tag), recursive=True):
        for r in csv.DictReader(open(f)):
            if "wino32" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                v += float(r["Counter_Value"]); n += 1
    return v, n
def dur(sub):
    t = n = 0
    for f in glob.glob("%s/%s_%s/**/*kernel_trace.csv" % (out, sub, tag), recursive=True):
        for r in csv.DictReader(open(f)):
            if "wino32" in r["Kernel_Name"]:
                t += float(r["End_Timestamp"]) - float(r["Start_Timestamp"]); n += 1
    return t / max(n, 1) / 1e6
f, nf = total("f", "FETCH_SIZE"); w, nw = total("w", "WRITE_SIZE")
B, H, C, K = [int(v) for v in shape.split()]
alg = (B * H * H * C + B * (H - 2) ** 2 * K + 16 * C * K) * 4 / 1e6
mb = 2 * f / max(nf, 1) / 1024 + w / max(nw, 1) / 1024
print("shape %-18s gn %-3s  %.3f ms  fetch x2 %.1f MB + write %.1f MB = %.1f MB per launch; algorithmic (U as read) %.1f MB -> %.2fx" %
      (shape, gn, dur("f"), 2 * f / max(nf, 1) / 1024, w / max(nw, 1) / 1024, mb, alg, mb / alg), flush=True)
PY
done
