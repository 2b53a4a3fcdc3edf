"""aggregate rocprofv3 counter_collection CSVs: mean counter value per kernel name"""
import csv
import glob
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
files = [f for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True) if len(sys.argv) < 3 or sys.argv[2] in f]
if files:
    print('columns:', open(files[0]).readline().strip())
for f in files:
    for row in csv.DictReader(open(f)):
        name = row['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
        acc[name][row['Counter_Name']].append(float(row['Counter_Value']))
for name in sorted(acc):
    if not any(k in name for k in ('k_ugemm', 'k_rowgemm', 'k_edge', 'k_node', 'k_rowsum', 'k_energy', 'k_sd_')):
        continue
    print(name)
    for c in sorted(acc[name]):
        v = acc[name][c]
        print('   %-34s mean %16.1f  (n=%d)' % (c, sum(v) / len(v), len(v)))
