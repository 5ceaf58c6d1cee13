import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"total {tot/1e6:.1f} ms  ({tot/1e6/steps:.2f} ms/step over {steps} steps)")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 26]:
    print(f"{float(r['TotalDurationNs'])/1e6/steps:8.2f} ms/step {float(r['Percentage']):6.2f}% n={int(r['Calls'])/steps:7.1f} avg={float(r['AverageNs'])/1e3:9.1f}us  {r['Name'][:100]}")
