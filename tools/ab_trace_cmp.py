#!/usr/bin/env python3
"""Per-launch durations (median over the last 10 forwards) of every library traced by tools/ab_trace.sh."""
import csv, glob, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D = os.path.join(ROOT, "gpurun_out", "abtrace")


def forward(tag):
    f = glob.glob(os.path.join(D, tag, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    st = [i for i, r in enumerate(rows) if "stem" in r["Kernel_Name"]]
    n = st[-1] - st[-2]
    fw = [rows[st[k]:st[k + 1]] for k in range(len(st) - 11, len(st) - 1) if st[k + 1] - st[k] == n]
    out = []
    for i in range(n):
        d = sorted((int(f_[i]["End_Timestamp"]) - int(f_[i]["Start_Timestamp"])) / 1e3 for f_ in fw)
        out.append((fw[-1][i]["Kernel_Name"].split("(")[0].replace("void bnn::", "")[:60], d[len(d) // 2]))
    return out


tags = ["main"] + sorted(t for t in os.listdir(D) if os.path.isdir(os.path.join(D, t)) and t != "main")
fw = {t: forward(t) for t in tags}
print("%-62s" % "", "".join("%9s" % t for t in tags))
for i, (name, _) in enumerate(fw["main"]):
    print("%-62s" % name, "".join("%9.1f" % fw[t][i][1] for t in tags))
print("%-62s" % "sum (us)", "".join("%9.0f" % sum(d for _, d in fw[t]) for t in tags))
