#!/usr/bin/env python3
"""Condenses gpurun_out/<tag>/ (tools/profile_round.sh) into profiles/<tag>_*: run locally after gpurun."""
import collections, csv, glob, json, os, shutil, sys
tag = sys.argv[1]
math_mode = sys.argv[2] if len(sys.argv) > 2 else "3"          # arithmetic mode the profiled run used (bench.py --math)
src = os.path.join("gpurun_out", tag)
os.makedirs("profiles", exist_ok=True)
bench = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
json.dump(bench, open("profiles/%s_bench.json" % tag, "w"), indent=1)
shutil.copy(glob.glob(os.path.join(src, "stats/*/*kernel_stats.csv"))[0], "profiles/%s_rocprofv3_kernel_stats.csv" % tag)
shutil.copy(os.path.join(src, "launches.csv"), "profiles/%s_launches.csv" % tag)
if os.path.exists(os.path.join(src, "bench_detail.json")):          # per-family / per-layer tables of the same run (bench.py --detail)
    shutil.copy(os.path.join(src, "bench_detail.json"), "profiles/%s_bench_detail.json" % tag)

def short(n):
    return n.split("(")[0].replace("void ", "").replace("unet::", "")[:44]

stats = list(csv.DictReader(open("profiles/%s_rocprofv3_kernel_stats.csv" % tag)))
DOM = (("wino", "wino32"), ("igemm", "igemm"), ("wgrad", "wgrad"))     # kernel families whose traffic / launch time are reported
EXCL = ("wino_transform", "reduce")
calls = {}; ns = {}
for key, sub in DOM:
    rs = [r for r in stats if sub in r["Name"] and not any(e in r["Name"] for e in EXCL)]
    calls[key] = sum(int(r["Calls"]) for r in rs); ns[key] = sum(float(r["TotalDurationNs"]) for r in rs)

def counters(sub):
    d = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen = set()
    f = glob.glob(os.path.join(src, sub, "*/*counter_collection.csv"))[0]
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"]); d[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (k, r["Dispatch_Id"]) not in seen:
            seen.add((k, r["Dispatch_Id"])); n[k] += 1
    dur = collections.defaultdict(float)
    f = glob.glob(os.path.join(src, sub, "*/*kernel_trace.csv"))[0]
    for r in csv.DictReader(open(f)):
        dur[short(r["Kernel_Name"])] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return d, n, dur

sq, nsq, dur = counters("sq"); fe, nfe, _ = counters("fetch"); wr, nwr, _ = counters("write"); ld, nld, _ = counters("lds")
lines = ["# %s — rocprofv3 PMC summary (bench.py --steps 2 --warmup 1, separate --pmc passes)" % tag, "",
         "FETCH_SIZE is doubled (gfx950 reports half of a wide coalesced read, MI355X_MICROARCH.md §HBM); units MB.",
         "MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x duration x clock); clock = GRBM_GUI_ACTIVE / 8 / duration.",
         "LDS conflict ratio = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (extra cycles / all LDS-array cycles, MI355X_MICROARCH.md LDS section).", "",
         "| kernel | launches | ms total | clock GHz | MFMA busy | WAIT_ANY/WAVE | fetch MB/launch (x2) | write MB/launch | LDS conflict ratio |",
         "|---|---|---|---|---|---|---|---|---|"]
tot_f = collections.Counter(); tot_w = collections.Counter(); tot_n = collections.Counter()
for k in sorted(dur, key=lambda k: -dur[k])[:24]:
    c = sq[k]; clk = c["GRBM_GUI_ACTIVE"] / 8 / dur[k] if dur[k] else 0
    busy = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * dur[k] * clk) if clk else 0
    f = 2 * fe[k]["FETCH_SIZE"] / 1024 / max(nfe[k], 1); w = wr[k]["WRITE_SIZE"] / 1024 / max(nwr[k], 1)
    lb = ld[k]["SQ_LDS_BANK_CONFLICT"] / max(ld[k]["SQ_LDS_IDX_ACTIVE"], 1)
    lines.append("| %s | %d | %.2f | %.2f | %.2f | %.2f | %.1f | %.1f | %.3f |" % (k, nsq[k], dur[k] / 1e6, clk, busy,
                 c["SQ_WAIT_ANY"] / max(c["SQ_WAVE_CYCLES"], 1), f, w, lb))
    for key, sub in DOM:
        if sub in k and not any(e in k for e in EXCL):
            tot_f[key] += 2 * fe[k]["FETCH_SIZE"] / 1024; tot_w[key] += wr[k]["WRITE_SIZE"] / 1024; tot_n[key] += nfe[k]
out = {"source": "profiles/%s_pmc_summary.md" % tag}
lines.append("")
bench_dom = "wino" if "wino" in bench["roofline"]["kernel"] else "igemm" if "igemm" in bench["roofline"]["kernel"] else "wgrad"
for key, sub in DOM:
    if not tot_n[key]:
        continue
    traffic = (tot_f[key] + tot_w[key]) / tot_n[key]
    out["%s_hbm_mb_per_launch" % key] = traffic
    lines.append("%s (all instantiations): HBM traffic %.1f MB per launch (fetch %.1f + write %.1f), %d launches." %
                 (sub, traffic, tot_f[key] / tot_n[key], tot_w[key] / tot_n[key], tot_n[key]))
    if calls[key]:
        lines.append("rocprofv3 --stats (bench.py --steps 5 --warmup 2): %s avg launch %.4f ms over %d calls%s." %
                     (sub, ns[key] / calls[key] / 1e6, calls[key],
                      "; bench.py HIP events: %.4f ms" % bench["roofline"]["avg_launch_ms"] if key == bench_dom else ""))
open("profiles/%s_pmc_summary.md" % tag, "w").write("\n".join(lines) + "\n")
allm = {}
if os.path.exists("profiles/pmc_traffic.json"):
    try:
        allm = json.load(open("profiles/pmc_traffic.json"))
    except ValueError:
        allm = {}
allm = {k: v for k, v in allm.items() if k.startswith("math")}
allm["math" + math_mode] = out
json.dump(allm, open("profiles/pmc_traffic.json", "w"), indent=1)
print("\n".join(lines[-5:]))
