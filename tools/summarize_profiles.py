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
sys.path.insert(0, os.getcwd())
import bench as _bench                      # repo root: the same fingerprint bench.py checks before quoting the traffic
out = {"source": "profiles/%s_pmc_summary.md" % tag, "csrc_sha": _bench.csrc_sha()}
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
# ---- per-layer HBM traffic of the dominant MFMA families: counter bytes against algorithmic bytes -------------------
# The PMC passes run whole steps (no instrumented pass), so the i-th dispatch of a kernel family within a step is the i-th
# launch of that family in bench.py's per-launch table (same launch order every step).
def per_layer_traffic(sub_kernel, kind_code, tagsub):
    def series(passdir, counter):
        f = glob.glob(os.path.join(src, passdir, "*/*counter_collection.csv"))[0]
        seen = collections.OrderedDict()
        for r in csv.DictReader(open(f)):
            if sub_kernel in r["Kernel_Name"] and "transform" not in r["Kernel_Name"] and "reduce" not in r["Kernel_Name"] and r["Counter_Name"] == counter:
                seen[int(r["Dispatch_Id"])] = seen.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
        return [seen[k] for k in sorted(seen)]
    launches = [r for r in csv.DictReader(open(os.path.join(src, "launches.csv"))) if int(r["kind"]) == kind_code and tagsub in r["tag"]]
    fe_s, wr_s = series("fetch", "FETCH_SIZE"), series("write", "WRITE_SIZE")
    if not launches or not fe_s:
        return []
    steps_in_table = 3 if len(launches) % 3 == 0 else 1
    per_step = len(launches) // steps_in_table
    if per_step == 0 or len(fe_s) % per_step or len(wr_s) != len(fe_s):
        return []
    rows = []
    for i in range(per_step):
        l = launches[i]
        f = 2 * sum(fe_s[i::per_step]) / (len(fe_s) // per_step) / 1024          # KB -> MB, x2 (gfx950 FETCH_SIZE)
        w = sum(wr_s[i::per_step]) / (len(wr_s) // per_step) / 1024
        alg = float(l["mbytes"])
        extra = ""
        if "tiles=" in l["tag"] and " N=" in l["tag"]:
            tiles = int(l["tag"].split("tiles=")[1].split()[0]); n = int(l["tag"].split(" N=")[1].split()[0])
            wgs = -(-tiles // 64) * (n // 32)
            extra = " | %d | %.2f" % (wgs, wgs / 512.0)
        rows.append("| %s | %.3f | %.1f | %.1f | %.2f%s |" % (l["row"], float(l["ms"]), alg, f + w, (f + w) / alg if alg else 0.0, extra))
    return rows

for title, subk, kind_code, tagsub, hdr in (("wino32_f32_kernel", "wino32", 3, "wino32", " | workgroups | rounds of 512"), ("igemmb_kernel", "igemmb", 0, "igemmb", "")):
    rows = per_layer_traffic(subk, kind_code, tagsub)
    if rows:
        lines += ["", "## %s: HBM bytes per launch, counters (FETCH_SIZE x2 + WRITE_SIZE) against algorithmic (SURVEY 8d)" % title, "",
                  "| launch | ms | algorithmic MB | counter MB | counter / algorithmic%s |" % hdr, "|---|---|---|---|---|" + ("---|---|" if hdr else "")] + rows
open("profiles/%s_pmc_summary.md" % tag, "w").write("\n".join(lines) + "\n")
allm = {}
if os.path.exists("profiles/pmc_traffic.json"):
    try:
        allm = json.load(open("profiles/pmc_traffic.json"))
    except ValueError:
        allm = {}
allm = {k: v for k, v in allm.items() if k.startswith("math")}
allm["math" + math_mode] = out
# The stamp is the fingerprint of dl-unet_amd/csrc at the time THIS script runs: run it on the tree the profile was taken on
# (right after the gpurun call), before editing any kernel.
json.dump(allm, open("profiles/pmc_traffic.json", "w"), indent=1)
print("\n".join(lines[-5:]))
