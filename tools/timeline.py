#!/usr/bin/env python3
"""developer tool: start / end of every kernel of ONE step from a rocprofv3 --kernel-trace csv (microseconds from the step's
first kernel), to see what ran beside what.  tools/timeline.py <kernel_trace.csv> [step index from the end, default 3]"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [k for k, r in enumerate(rows) if "k_world_step" in r["Kernel_Name"]]
a = starts[-back - 1]; b = starts[-back]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    m = re.search(r"(k_[a-z_]+(<[^>]*>)?)", r["Kernel_Name"]); name = m.group(1) if m else r["Kernel_Name"][:30]
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print("%-26s %9.1f -> %9.1f  (%7.1f us)  queue %s" % (name, s, e, e - s, r.get("Queue_Id", "?")))
# averages per (kernel, occurrence within the step) over the last 40 steps
import collections
acc = collections.defaultdict(list)
for a, b in zip(starts[-41:-1], starts[-40:]):
    seen = collections.Counter()
    t0 = int(rows[a]["Start_Timestamp"])
    for r in rows[a:b]:
        m = re.search(r"(k_[a-z_]+)", r["Kernel_Name"]); name = m.group(1) if m else r["Kernel_Name"][:30]
        seen[name] += 1
        acc[(name, seen[name])].append(((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3))
print("-- averages over 40 steps: kernel, occurrence, start, end, duration")
for (name, occ), v in sorted(acc.items(), key=lambda kv: sum(x[0] for x in kv[1]) / len(kv[1])):
    s = sum(x[0] for x in v) / len(v); e = sum(x[1] for x in v) / len(v)
    print("%-22s #%d  %8.1f -> %8.1f  (%7.1f)  n=%d" % (name, occ, s, e, e - s, len(v)))
