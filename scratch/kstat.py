import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r['Name']:
        print(f"  {r['Name'][:58]:58s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us")
