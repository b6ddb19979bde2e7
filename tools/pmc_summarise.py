"""Per-kernel average of FETCH_SIZE / WRITE_SIZE (KB per launch) from two rocprofv3 --pmc passes."""
import csv
import glob
import json
import sys
from collections import defaultdict

out = defaultdict(dict)
for d, name in ((sys.argv[1], 'FETCH_SIZE'), (sys.argv[2], 'WRITE_SIZE')):
    acc = defaultdict(list)
    for path in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(path)):
            if row.get('Counter_Name') == name:
                acc[row['Kernel_Name']].append(float(row['Counter_Value']))
    for k, v in acc.items():
        if 'fnn::' in k or 'k_' in k:
            out[k[:60]][name + '_KB_per_launch'] = sum(v) / len(v)
            out[k[:60]]['launches'] = len(v)
print(json.dumps(out, indent=1))
