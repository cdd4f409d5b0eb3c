"""Timeline of the last run in a rocprofv3 kernel trace: start and end of every dispatch relative to the first."""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
per = len(rows) // runs
last = rows[-per:]
t0 = int(last[0]['Start_Timestamp'])
for r in last:
    name = re.sub(r'\(.*', '', r['Kernel_Name'].replace('sa::(anonymous namespace)::', '').replace('void ', ''))
    s, e = (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3
    g = [int(r['Grid_Size_' + a]) // int(r['Workgroup_Size_' + a]) for a in 'XYZ']
    print('%8.1f %8.1f  %6.1f us  q%-3s %-22s blocks %s' % (s, e, e - s, r.get('Queue_Id', '?'), name, g))
