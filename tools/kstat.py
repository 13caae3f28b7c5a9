import csv,glob,sys
f=glob.glob('/tmp/kt/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r['Name']
    if n.startswith('k_') or 'k_lift' in n:
        print(f"{n[:28]:30s} calls {r['Calls']:>4s} avg_us {float(r['AverageNs'])/1e3:9.1f}")
