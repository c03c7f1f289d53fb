"""Aggregate rocprofv3 counter_collection csv files by (kernel, counter): mean value per dispatch.
usage: pmc_agg.py <dir> [out.json]   (text table on stdout, optional JSON {kernel: {counter: mean, n: launches}})"""
import csv, sys, glob, collections, re, json
acc = collections.defaultdict(lambda: [0.0, 0])
for path in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            k = re.sub(r'\(.*', '', r['Kernel_Name'])
            k = re.sub(r'^void ', '', k)[:90]
            a = acc[(k, r['Counter_Name'])]
            a[0] += float(r['Counter_Value']); a[1] += 1
out = collections.defaultdict(dict)
for (k, c), (s, n) in sorted(acc.items()):
    print(f'{k:90s} {c:28s} n={n:5d} mean={s / n:.6g}')
    out[k][c] = s / n
    out[k]['n'] = n
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], 'w'), indent=1, sort_keys=True)
