#!/bin/bash
# Socket power and engine clock while the stem kernel runs back to back (rocm-smi every 250 ms).
R="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$R"
( while true; do rocm-smi -c -P --json 2>/dev/null | tr -d '\n'; echo; sleep 0.25; done ) > /tmp/smi_stem.jsonl &
SP=$!
timeout 60 python tools/exp_stem_loop.py
kill $SP 2>/dev/null; wait $SP 2>/dev/null
python - <<'PY'
import json, re
rows = []
for ln in open("/tmp/smi_stem.jsonl"):
    try: rows.append(json.loads(ln).get("card0", {}))
    except Exception: pass
for k in [k for k in rows[0] if "sclk" in k.lower() or "power" in k.lower()]:
    vals = []
    for r in rows:
        m = re.search(r"[-+]?\d*\.?\d+", str(r.get(k, "")).replace("(", " "))
        if m: vals.append(float(m.group()))
    if vals: print("%-45s n=%3d  min %8.1f  median %8.1f  max %8.1f" % (k, len(vals), min(vals), sorted(vals)[len(vals)//2], max(vals)))
PY
