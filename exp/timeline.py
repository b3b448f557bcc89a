"""print the per-dispatch timeline of the last bench step from a rocprofv3 --kernel-trace csv"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if 'h2e_' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last step = after the last big gap: take the last N dispatches where N = dispatches per step (count of kernels / steps)
n_steps = int(sys.argv[2])
per = len(rows) // n_steps
last = rows[-per:]
t0 = int(last[0]['Start_Timestamp'])
for r in last:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].replace('void ', '')[:34]
    print(f"{(s-t0)/1e6:8.3f} {(e-t0)/1e6:8.3f} {(e-s)/1e6:8.3f}  grid={r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size')} q={r.get('Queue_Id')} {name}")
