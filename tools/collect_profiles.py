#!/usr/bin/env python3
"""Turn the scratch output of tools/gpu_profile.sh (gpurun_out/final/) into the committed summaries:
   profiles/<tag>_bench.json (+ _c2 / _c5), <tag>_bench_kernel_stats.csv (two batches in flight: the default bench
   command) and <tag>_bench_kernel_stats_1stream.csv (one batch in flight: per-kernel times without overlap),
   <tag>_c2_pmc_counters.json, <tag>_stem_pmc.json.
Usage (in the build container, after the gpurun call):  python tools/collect_profiles.py r01"""
import collections, csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
import subprocess
# the GPU box runs a snapshot of the working tree: collect right after the run, from a clean tree, and stamp the commit
COMMIT = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip()
DIRTY = bool(subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--untracked-files=no"], capture_output=True,
                            text=True).stdout.strip())
STAMP = {"commit": COMMIT + ("+uncommitted" if DIRTY else ""), "collected_by": "tools/gpu_profile.sh + tools/collect_profiles.py"}
src = os.path.join(ROOT, "gpurun_out", "final")
# gpurun MERGES what a visit wrote into the local gpurun_out/: files of an earlier visit stay.  Remove the local
# gpurun_out/final BEFORE the visit (tools/gpu_profile.sh removes it on the GPU side only), or summaries of an old build
# get this commit's stamp.  Guard: everything that is summarised must be younger than the bench line of this visit.
_t0 = os.path.getmtime(os.path.join(src, "bench.json")) - 3600


def fresh(paths):
    return [f for f in paths if os.path.getmtime(f) >= _t0]

dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)

def stamped(line, cmd):
    rec = json.loads(line)
    rec["provenance"] = dict(STAMP, command=cmd)
    return json.dumps(rec)


line = open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1]
open(os.path.join(dst, f"{tag}_bench.json"), "w").write(stamped(line, "python bench.py --steps 20 --warmup 5") + "\n")
for extra in ("c2", "c5"):
    fn = os.path.join(src, f"bench_{extra}.json")
    if os.path.exists(fn) and open(fn).read().strip():
        ln = open(fn).read().strip().splitlines()[-1]
        cmd = {"c2": "python bench.py --config c2 --steps 20 --warmup 5",
               "c5": "python bench.py --config c5 --no-cpu-baseline --no-roofline"}[extra]
        open(os.path.join(dst, f"{tag}_bench_{extra}.json"), "w").write(stamped(ln, cmd) + "\n")
stats = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(dst, f"{tag}_bench_kernel_stats.csv"))
stats1 = glob.glob(os.path.join(src, "stats1", "**", "*kernel_stats.csv"), recursive=True)
if stats1:
    shutil.copy(stats1[0], os.path.join(dst, f"{tag}_bench_kernel_stats_1stream.csv"))

for name in ("train.txt", "stem_ab.txt", "grad_pmc.txt", "train_last_step.txt"):
    fn = os.path.join(src, name)
    if os.path.exists(fn) and fresh([fn]) and open(fn).read().strip():
        body = "\n".join(l for l in open(fn).read().splitlines() if "amdgpu.ids" not in l)
        open(os.path.join(dst, f"{tag}_{name}"), "w").write(
            f"# {name}: tools/gpu_profile.sh, commit {STAMP['commit']}\n" + body + "\n")

for f in fresh(glob.glob(os.path.join(src, "roctx", "**", "*marker_api_stats.csv"), recursive=True))[:1]:
    shutil.copy(f, os.path.join(dst, f"{tag}_roctx_marker_stats.csv"))

# stem kernel: counters of tools/bench_stem.py (ONLY=default), separate passes
sagg = collections.defaultdict(list)
sdur = []
for f in fresh(glob.glob(os.path.join(src, "stem_*", "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        if "stem_rows_kernel" in r["Kernel_Name"]:
            sagg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in fresh(glob.glob(os.path.join(src, "stem_a", "**", "*kernel_trace.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        if "stem_rows_kernel" in r["Kernel_Name"]:
            sdur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
if sagg:
    stem = {c: int(round(sum(v) / len(v))) for c, v in sorted(sagg.items())}
    if sdur:
        stem["avg_duration_us_profiled"] = round(sum(sdur) / len(sdur), 1)
    stem["kernel"] = "bnn::stem_rows_kernel<false, false>, batch 256, 224x224, fp32 + sign planes out (tools/bench_stem.py)"
    stem["provenance"] = STAMP
    json.dump(stem, open(os.path.join(dst, f"{tag}_stem_pmc.json"), "w"), indent=1, sort_keys=True)

agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in [g for g in fresh(glob.glob(os.path.join(src, "*", "**", "*counter_collection.csv"), recursive=True))
          if os.sep + "stem_" not in g]:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0]
        if "bnn::" in name:
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in fresh(glob.glob(os.path.join(src, "sq1", "**", "*kernel_trace.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0]
        if "bnn::" in name:
            dur[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = {}
for name, d in agg.items():
    out[name] = {c: int(round(sum(v) / len(v))) for c, v in sorted(d.items())}
    if dur[name]:
        out[name]["avg_duration_us_profiled"] = round(sum(dur[name]) / len(dur[name]), 1)
if out:      # (QUICK visits have no PMC passes: write nothing rather than an empty summary)
    out["provenance"] = STAMP
    json.dump(out, open(os.path.join(dst, f"{tag}_c2_pmc_counters.json"), "w"), indent=1, sort_keys=True)
    del out["provenance"]

# the one-launch layer on config 2 (tools/run_fly.py under rocprofv3, separate passes)
fagg = collections.defaultdict(list)
fdur = []
for f in fresh(glob.glob(os.path.join(src, "fly_*", "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        if "bconv_fly" in r["Kernel_Name"]:
            fagg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in fresh(glob.glob(os.path.join(src, "fly_sq1", "**", "*kernel_trace.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        if "bconv_fly" in r["Kernel_Name"]:
            fdur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
if fagg:
    fly = {c: int(round(sum(v) / len(v))) for c, v in sorted(fagg.items())}
    if fdur:
        fly["avg_duration_us_profiled"] = round(sum(fdur) / len(fdur), 1)
    fly["kernel"] = ("bnn::bconv_fly_kernel<3,3,4,false,false>: Conv2d 128->128 3x3 pad 1, x [256,128,56,56] fp32 in, "
                     "fp32 out, default plan (tools/run_fly.py)")
    if "FETCH_SIZE" in fly and "WRITE_SIZE" in fly:
        fly["hbm_bytes_per_launch"] = (2 * fly["FETCH_SIZE"] + fly["WRITE_SIZE"]) * 1024
        fly["algorithmic_bytes"] = 256 * 128 * 56 * 56 * 4 * 2 + 128 * 1152 // 8
        fly["traffic_over_algorithmic"] = round(fly["hbm_bytes_per_launch"] / fly["algorithmic_bytes"], 3)
        fly["note"] = "FETCH_SIZE / WRITE_SIZE in KiB; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950"
    fly["provenance"] = STAMP
    json.dump(fly, open(os.path.join(dst, f"{tag}_c2_fused_pmc.json"), "w"), indent=1, sort_keys=True)
    print("fly", {k: fly.get(k) for k in ("SQ_INSTS_VALU", "SQ_WAVES", "avg_duration_us_profiled", "traffic_over_algorithmic")})
stats_c2 = glob.glob(os.path.join(src, "stats_c2", "**", "*kernel_stats.csv"), recursive=True)
if stats_c2:
    shutil.copy(stats_c2[0], os.path.join(dst, f"{tag}_bench_c2_kernel_stats.csv"))
stats_c2_1 = glob.glob(os.path.join(src, "stats_c2_1", "**", "*kernel_stats.csv"), recursive=True)
if stats_c2_1:      # one launch at a time, nothing else in the process: the clean average of bconv_fly_kernel
    shutil.copy(stats_c2_1[0], os.path.join(dst, f"{tag}_bench_c2_kernel_stats_1stream.csv"))
    ln = [l for l in open(os.path.join(src, "stats_c2_1.log")).read().splitlines() if l.startswith("{")]
    if ln:
        open(os.path.join(dst, f"{tag}_bench_c2_1stream.json"), "w").write(stamped(
            ln[-1], "rocprofv3 --kernel-trace --stats -- python bench.py --config c2 --steps 20 --warmup 5 --sustain 0 "
                    "--no-extras --no-roofline --no-cpu-baseline") + "\n")
for n in ("layerwise_library", "layerwise", "fused", "fused_exact_stem"):     # written by tests/test_gpu_c3_full.py on the GPU box
    fn = os.path.join(ROOT, "gpurun_out", f"c3_b256_parity_{n}.json")
    if os.path.exists(fn):
        rec = json.load(open(fn))
        rec["provenance"] = STAMP
        json.dump(rec, open(os.path.join(dst, f"{tag}_c3_b256_parity_{n}.json"), "w"), indent=1)
fn = os.path.join(ROOT, "gpurun_out", "c5_b32_parity.json")      # tests/test_gpu_c3_full.py::test_c5_hblock_3463_at_its_stated_size
if os.path.exists(fn) and fresh([fn]):
    rec = json.load(open(fn))
    rec["provenance"] = STAMP
    json.dump(rec, open(os.path.join(dst, f"{tag}_c5_b32_parity.json"), "w"), indent=1)
# clocks / power under load, two-stream timeline (tools/power_and_overlap.sh)
for name, out_name in (("overlap.txt", "two_stream_overlap.txt"), ("summary.txt", "clock_power_under_load.txt")):
    fn = os.path.join(ROOT, "gpurun_out", "power", name)
    if os.path.exists(fn) and fresh([fn]) and open(fn).read().strip():
        open(os.path.join(dst, f"{tag}_{out_name}"), "w").write(
            f"# tools/power_and_overlap.sh, commit {STAMP['commit']}\n" + open(fn).read())
print("wrote", sorted(os.listdir(dst)))
for name, d in out.items():
    print(name[:70], {k: d[k] for k in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_WAVES", "avg_duration_us_profiled") if k in d})
