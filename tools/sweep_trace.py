import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'sweep_' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
k = int(sys.argv[2]) if len(sys.argv) > 2 else 12
print([round((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3) for r in rows[-k:]])
