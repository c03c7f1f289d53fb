"""Aggregate a rocprofv3 counter_collection csv by (kernel, counter): mean value per dispatch."""
import csv, sys, glob, collections, re
acc = collections.defaultdict(lambda: [0.0, 0])
for path in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            k = re.sub(r'\(.*', '', r['Kernel_Name'])[:70]
            a = acc[(k, r['Counter_Name'])]
            a[0] += float(r['Counter_Value']); a[1] += 1
for (k, c), (s, n) in sorted(acc.items()):
    print(f'{k:70s} {c:28s} n={n:5d} mean={s / n:.4g}')
