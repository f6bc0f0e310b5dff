import csv, collections, sys, glob
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name']
    if len(sys.argv) > 2 and sys.argv[2] not in n: continue
    key = (n[:60], int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1))
    agg[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = sum(sum(v) for v in agg.values())
print("total ms %.2f" % (tot / 1e3))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:int(sys.argv[3]) if len(sys.argv) > 3 else 25]:
    print(f"{k[0]:60s} blocks {k[1]:6d} calls {len(v):5d} avg_us {sum(v)/len(v):8.1f} tot_ms {sum(v)/1e3:8.2f}")
