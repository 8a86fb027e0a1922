"""Kernel time of the LAST step in a rocprofv3 kernel trace:
   python tools/last_step_profile.py trace.csv <step_ms> [anchor]     (window ends with the last kernel whose name has `anchor`)"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
win = float(sys.argv[2]) * 1e6
anchor = sys.argv[3] if len(sys.argv) > 3 else ""
end = max(int(r["End_Timestamp"]) for r in rows if anchor in r["Kernel_Name"])
sel = [r for r in rows if end - win <= int(r["Start_Timestamp"]) and int(r["End_Timestamp"]) <= end]
agg = collections.defaultdict(lambda: [0, 0])
for r in sel:
    n = r["Kernel_Name"]
    n = n.replace("void ", "").replace("bnn::", "").replace("(anonymous namespace)::", "").split("(")[0].replace("at::native::", "")[:110]
    agg[n][0] += 1
    agg[n][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(v[1] for v in agg.values())
print("kernels in the last %.1f ms: %d, busy %.2f ms" % (win / 1e6, len(sel), tot / 1e6))
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:48]:
    print("%8.3f ms %5d  %s" % (t / 1e6, c, n))
