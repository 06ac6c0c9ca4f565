import csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last call: find last walk_pipe kernel group; print the last 14 kernels with start offsets
last = rows[-14:]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us +%8.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:60]))
