import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
dur = collections.defaultdict(float)
seen = set()
for r in rows:
    k = r['Kernel_Name'].split('(')[0][-40:]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    disp[k].add(r['Dispatch_Id'])
    if r['Dispatch_Id'] not in seen:
        seen.add(r['Dispatch_Id'])
        dur[k] += (int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, v in agg.items():
    if 'rtamd' not in k: continue
    print(k, 'dispatches', len(disp[k]), 'time_ms %.3f' % (dur[k] / 1e6))
    for c, x in sorted(v.items()):
        print('    %-28s %.4g' % (c, x))
