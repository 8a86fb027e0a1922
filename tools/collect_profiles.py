#!/usr/bin/env python3
"""Turn the scratch output of tools/gpu_profile.sh (gpurun_out/final/) into the committed summaries:
   profiles/<tag>_bench.json (+ _c2 / _c5), <tag>_bench_kernel_stats.csv (two batches in flight: the default bench
   command) and <tag>_bench_kernel_stats_1stream.csv (one batch in flight: per-kernel times without overlap),
   <tag>_c2_pmc_counters.json, <tag>_stem_pmc.json.
Usage (in the build container, after the gpurun call):  python tools/collect_profiles.py r01"""
import collections, csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", "final")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)

line = open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1]
json.loads(line)
open(os.path.join(dst, f"{tag}_bench.json"), "w").write(line + "\n")
for extra in ("c2", "c5"):
    fn = os.path.join(src, f"bench_{extra}.json")
    if os.path.exists(fn) and open(fn).read().strip():
        ln = open(fn).read().strip().splitlines()[-1]
        json.loads(ln)
        open(os.path.join(dst, f"{tag}_bench_{extra}.json"), "w").write(ln + "\n")
stats = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(dst, f"{tag}_bench_kernel_stats.csv"))
stats1 = glob.glob(os.path.join(src, "stats1", "**", "*kernel_stats.csv"), recursive=True)
if stats1:
    shutil.copy(stats1[0], os.path.join(dst, f"{tag}_bench_kernel_stats_1stream.csv"))

# stem kernel: counters of tools/bench_stem.py (ONLY=default), separate passes
sagg = collections.defaultdict(list)
sdur = []
for f in glob.glob(os.path.join(src, "stem_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "stem_split_kernel" in r["Kernel_Name"]:
            sagg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(os.path.join(src, "stem_a", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "stem_split_kernel" in r["Kernel_Name"]:
            sdur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
if sagg:
    stem = {c: int(round(sum(v) / len(v))) for c, v in sorted(sagg.items())}
    if sdur:
        stem["avg_duration_us_profiled"] = round(sum(sdur) / len(sdur), 1)
    stem["kernel"] = "bnn::stem_split_kernel<false>, batch 256, 224x224, fp32 + sign planes out (tools/bench_stem.py)"
    json.dump(stem, open(os.path.join(dst, f"{tag}_stem_pmc.json"), "w"), indent=1, sort_keys=True)

agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in [g for g in glob.glob(os.path.join(src, "*", "**", "*counter_collection.csv"), recursive=True)
          if os.sep + "stem_" not in g]:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0]
        if "bnn::" in name:
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(os.path.join(src, "sq1", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0]
        if "bnn::" in name:
            dur[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = {}
for name, d in agg.items():
    out[name] = {c: int(round(sum(v) / len(v))) for c, v in sorted(d.items())}
    if dur[name]:
        out[name]["avg_duration_us_profiled"] = round(sum(dur[name]) / len(dur[name]), 1)
json.dump(out, open(os.path.join(dst, f"{tag}_c2_pmc_counters.json"), "w"), indent=1, sort_keys=True)
print("wrote", sorted(os.listdir(dst)))
for name, d in out.items():
    print(name[:70], {k: d[k] for k in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_WAVES", "avg_duration_us_profiled") if k in d})
