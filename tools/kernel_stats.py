import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
n=float(sys.argv[2]) if len(sys.argv)>2 else 1
tot=sum(float(r['TotalDurationNs']) for r in rows)
print(f"total {tot/1e6/n:.3f} ms per update, {sum(int(r['Calls']) for r in rows)/n:.0f} launches")
for r in rows[:int(sys.argv[3]) if len(sys.argv)>3 else 50]:
    print(f"{r['Name'][:100]:100s} {int(r['Calls'])/n:7.1f} {float(r['TotalDurationNs'])/1e6/n:8.3f}ms {float(r['AverageNs'])/1e3:8.1f}us {float(r['Percentage']):5.1f}")
