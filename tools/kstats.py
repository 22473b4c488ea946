"""Prints the top rows of a rocprofv3 kernel_stats.csv: python tools/kstats.py <csv> [rows] [name filter]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
flt = sys.argv[3] if len(sys.argv) > 3 else ""
for r in [r for r in rows if flt in r["Name"]][:n]:
    print(f"{r['Name'][:64]:64s} calls={r['Calls']:>6s} avg={float(r['AverageNs']) / 1e3:9.1f} us total={float(r['TotalDurationNs']) / 1e6:9.2f} ms {float(r['Percentage']):5.1f}%")
