import csv, sys, re
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "tiny_step" in r["Kernel_Name"]]
# take a window in the middle of the tiny kernels
mid = idx[len(idx) // 2]
prev = None
for r in rows[mid - 2: mid + 14]:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")).replace("void ", "")[:40]
    print("%8.1f us gap %6.1f dur %6.1f  %s" % (0 if prev is None else (st - t0) / 1e3, 0 if prev is None else (st - prev) / 1e3, (en - st) / 1e3, name))
    if prev is None: t0 = st
    prev = en
