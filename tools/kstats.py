"""Print the csmpn / fill / copy rows of a rocprofv3 kernel_stats.csv (name shortened, calls, average ns)."""
import csv, glob, os, sys
for d in sys.argv[1:]:
    f = d if d.endswith('.csv') else (sorted(glob.glob(os.path.join(d, '**', '*kernel_stats.csv'), recursive=True)) or [None])[0]
    print('==', d)
    if not f:
        print('  no kernel_stats.csv'); continue
    for r in csv.DictReader(open(f)):
        n = r['Name']
        if not any(k in n for k in ('csmpn', 'Fill', 'copyBuffer', 'segment')): continue
        n = n.replace('csmpn::Alg<3, 0u>, ', '').replace('(csmpn::DevCemlp, csmpn::RowIO)', '').replace('void ', '').replace('csmpn::', '')
        print(f"  {n[:70]:70s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs'])/1e3:9.2f} us  total {float(r['TotalDurationNs'])/1e3:10.1f} us")
