# per-launch durations of the hysteresis passes from a rocprofv3 kernel trace csv
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
d = [(int(r['Start_Timestamp']), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3) for r in rows if 'hyst_pass' in r['Kernel_Name']]
d.sort()
v = [x[1] for x in d]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
print('last clip:', [round(x, 1) for x in v[-n:]], 'sum', round(sum(v[-n:]), 1))
