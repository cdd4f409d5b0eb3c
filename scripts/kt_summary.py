"""Per-dispatch listing of the last batch in a rocprofv3 kernel trace."""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
half = len(rows) // 2
for r in rows[half:]:
    name = re.sub(r'\(.*', '', r['Kernel_Name'].replace('sa::(anonymous namespace)::', '').replace('void ', ''))
    dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    g = [int(r['Grid_Size_' + a]) // int(r['Workgroup_Size_' + a]) for a in 'XYZ']
    print('%-22s blocks %-16s vgpr %-4s lds %-6s dur %8.1f us' % (name, g, r['VGPR_Count'], r['LDS_Block_Size'], dur))
