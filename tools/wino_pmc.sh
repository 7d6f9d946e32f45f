#!/bin/bash
# usage: tools/wino_pmc.sh <outdir-under-gpurun_out> [B H C K [mode]] ; PMC passes over one conv launch shape
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P="python3 $GRAFT_REPO_ROOT/tools/wino_one.py $*"
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/sq --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -- $P > $OUT/sq.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/lds --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM -- $P > $OUT/lds.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/misc --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM -- $P > $OUT/misc.log 2>&1 || true
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in sorted(glob.glob(out + "/*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        if "wino_f32" not in kn and "igemm_f32" not in kn: continue
        kn = kn.split("(")[0][:40]
        acc[kn][r["Counter_Name"]] += float(r["Counter_Value"])
    for kn, d in acc.items():
        print(f.split("/")[-3] if False else f.replace(out, ""), kn)
        for c, v in sorted(d.items()): print("   %-28s %.4g" % (c, v / 3))
PY
