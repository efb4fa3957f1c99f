"""Summarise rocprofv3 --pmc CSVs: per kernel, mean counter value per dispatch."""
import csv, glob, os, sys, collections, json
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, 'p*', '**', '*counter_collection.csv'), recursive=True):
    for row in csv.DictReader(open(f)):
        name = row['Kernel_Name'].split('(')[0].replace('(anonymous namespace)::', '')
        name = row['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0]
        acc[name][row['Counter_Name']].append(float(row['Counter_Value']))
res = {}
for k, d in acc.items():
    res[k] = {c: sum(v) / len(v) for c, v in d.items()}
    res[k]['_dispatches'] = max(len(v) for v in d.values())
json.dump(res, open(os.path.join(out, 'pmc_summary.json'), 'w'), indent=1, sort_keys=True)
for k in sorted(res):
    if 'ext' in k or 'transit' in k or 'kmax' in k or 'transmission' in k:
        print(k)
        for c in sorted(res[k]):
            print(f'   {c:40s} {res[k][c]:.4g}')
