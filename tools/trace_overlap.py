"""Timeline of a rocprofv3 --kernel-trace of `bench.py` with two batches in flight: which kernels run beside which.

    python tools/trace_overlap.py <kernel_trace.csv> [n_tail_kernels]

Prints, for the last part of the trace (steady state): the share of wall time with 0 / 1 / 2+ kernels in flight; per
kernel class (stem, conv, head) the average duration when it runs ALONE, beside a stem of the other queue, beside a
convolution of the other queue; and a short timeline."""
import csv, sys, collections

rows = list(csv.DictReader(open(sys.argv[1])))
col = lambda *keys: next(c for c in rows[0] if all(k.lower() in c.lower() for k in keys))
cn, cs, ce = col("kernel", "name"), col("start"), col("end")
try:
    cq = col("queue")
except StopIteration:
    cq = col("stream")
ev = []
for r in rows:
    n = r[cn]
    cls = "stem" if "stem_rows" in n else "conv" if "bconv_sgpr" in n else "head" if ("avgpool" in n or "fc_ws" in n) else None
    if cls:
        ev.append((int(r[cs]), int(r[ce]), cls, r[cq], n))
ev.sort()
tail = int(sys.argv[2]) if len(sys.argv) > 2 else 1200
ev = ev[-tail:]
t0, t1 = ev[0][0], max(e[1] for e in ev)
# wall-time share by number of kernels in flight
pts = sorted([(s, 1) for s, e, *_ in ev] + [(e, -1) for s, e, *_ in ev])
share, depth, last = collections.Counter(), 0, t0
for t, d in pts:
    share[min(depth, 2)] += t - last
    depth, last = depth + d, t
tot = sum(share.values())
print("wall %.1f us per forward-pair slot; kernels in flight: " % ((t1 - t0) / 1e3 / (sum(1 for e in ev if e[2] == "stem") or 1)) +
      ", ".join("%d: %.1f%%" % (k, 100 * v / tot) for k, v in sorted(share.items())))
queues = sorted({e[3] for e in ev})
print("queues", queues, "kernels", len(ev))
# per class: duration by what the other queue runs beside it
acc = collections.defaultdict(list)
for i, (s, e, cls, q, n) in enumerate(ev):
    ov = collections.Counter()
    for (s2, e2, cls2, q2, n2) in ev[max(0, i - 40): i + 40]:
        if q2 != q:
            o = min(e, e2) - max(s, s2)
            if o > 0:
                ov[cls2] += o
    d = e - s
    beside = "alone" if sum(ov.values()) < 0.2 * d else max(ov, key=ov.get)
    acc[(cls, beside)].append(d / 1e3)
    acc[(cls, "any")].append(d / 1e3)
for k in sorted(acc):
    v = acc[k]
    print("%-5s beside %-6s n=%4d  avg %7.1f us  sum/forward %7.1f us" % (k[0], k[1], len(v), sum(v) / len(v),
          sum(v) / max(1, len(acc[("stem", "any")]))))
print("-- timeline (last 60 kernels): start us, dur us, queue, class")
for s, e, cls, q, n in ev[-60:]:
    print("%9.1f %7.1f  q%-3s %s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, cls, n.split("<")[1][:40] if "<" in n else ""))
