import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0
for r in rows:
    c = int(r["Calls"])
    if c >= 100:
        tot += float(r["TotalDurationNs"]) / 101 / 1000
        print(f"{r['Name'].split('(')[0][-70:]:72s} per-iter {c/101:6.1f}  avg {float(r['AverageNs'])/1000:7.2f} us  per-iter-us {float(r['TotalDurationNs'])/101/1000:8.1f}")
print("sum of kernel time per iteration (us):", round(tot, 1))
