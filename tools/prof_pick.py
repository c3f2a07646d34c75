import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r['Name'] for k in sys.argv[2:]):
        print(f"  {r['Name'][11:60]:50s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:7.1f} us  min {int(r['MinNs'])/1e3:7.1f}")
