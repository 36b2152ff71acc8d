"""Aggregate a rocprofv3 counter-collection CSV per kernel: mean counter value per launch.

usage: python tests/tools/pmc_summary.py <dir with *_counter_collection.csv> <out.csv>
"""
import csv, glob, os, sys
from collections import defaultdict

def main(src, out):
    files = glob.glob(os.path.join(src, '**', '*counter_collection.csv'), recursive=True)
    if not files:
        raise SystemExit('no *counter_collection.csv under ' + src)
    acc = defaultdict(lambda: [0.0, 0])
    for f in files:
        for r in csv.DictReader(open(f)):
            key = (r['Kernel_Name'], r['Counter_Name'])
            acc[key][0] += float(r['Counter_Value'])
            acc[key][1] += 1
    with open(out, 'w', newline='') as fo:
        w = csv.writer(fo)
        w.writerow(['Kernel_Name', 'Counter_Name', 'Launches', 'MeanPerLaunch', 'Total'])
        for (k, c), (tot, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
            w.writerow([k[:160], c, n, '%.6g' % (tot / n), '%.6g' % tot])

if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
