"""Condenses a rocprofv3 --kernel-trace CSV into a per-(kernel, grid) table: calls, average / total duration.
usage: python tools/summarize_profile.py <kernel_trace.csv> [out.md]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", ""))
    name = re.sub(r"^void ", "", name)
    if len(name) > 70:
        name = name[:67] + "..."
    wg = int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1) * max(int(r["Grid_Size_Y"]), 1) * max(int(r["Grid_Size_Z"]), 1)
    agg[(name, wg)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in agg.values())
lines = ["| kernel | workgroups | calls | avg us | total ms | % |", "|---|---|---|---|---|---|"]
for (name, wg), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    if sum(v) / tot < 0.002:
        continue
    lines.append(f"| `{name}` | {wg} | {len(v)} | {sum(v) / len(v):.1f} | {sum(v) / 1e3:.2f} | {100 * sum(v) / tot:.1f} |")
lines.append(f"\ntotal kernel time {tot / 1e3:.2f} ms over {len(rows)} dispatches")
out = "\n".join(lines)
print(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")
