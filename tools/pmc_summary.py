import csv, sys, collections
for f in sys.argv[1:]:
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    seen=set()
    for r in rows:
        k = r['Kernel_Name'].split('(')[0][:60]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        key=(k, r['Dispatch_Id'])
        if key not in seen: seen.add(key); cnt[k]+=1
    for k,v in agg.items():
        print(k, 'dispatches', cnt[k])
        for c,val in sorted(v.items()): print('   %-24s %14.0f  per-dispatch %12.0f' % (c, val, val/cnt[k]))
