import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
by = collections.OrderedDict()
for r in rows:
    if 'h2e_' in r['Kernel_Name']:
        k = (int(r['Dispatch_Id']), r['Kernel_Name'].replace('void ', '')[:32], r['Grid_Size'],
             round((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6, 3))
        by.setdefault(k, {})[r['Counter_Name'].replace('SQ_', '')] = int(float(r['Counter_Value']))
ks = sorted(by)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
for k in ks[-n:]:
    print(k, by[k])
