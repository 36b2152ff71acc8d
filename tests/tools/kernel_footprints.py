"""LDS / register footprint of every kernel of a rocprofv3 --kernel-trace run (the CSV's LDS_Block_Size, VGPR_Count, SGPR_Count,
Workgroup_Size columns): which kernels can share a CU with a persistent 144-KB GEMM workgroup (16 KB of LDS and, with two or three
~200- / ~168-register waves per SIMD already there, under 100 registers left).  Written for RCCL's kernels in the forced one-rank
exchange.  usage: python tests/tools/kernel_footprints.py <trace dir>"""
import csv, glob, os, sys
from collections import defaultdict
files = glob.glob(os.path.join(sys.argv[1], '**', '*kernel_trace.csv'), recursive=True)
if not files:
    raise SystemExit('no *kernel_trace.csv under ' + sys.argv[1])
acc = defaultdict(lambda: [0, 0.0, None])
for f in files:
    for r in csv.DictReader(open(f)):
        name = r.get('Kernel_Name', '')
        key = name[:90]
        dur = (float(r['End_Timestamp']) - float(r['Start_Timestamp'])) * 1e-3
        a = acc[key]
        a[0] += 1; a[1] += dur
        a[2] = (r.get('LDS_Block_Size'), r.get('VGPR_Count'), r.get('Accum_VGPR_Count'), r.get('SGPR_Count'), r.get('Workgroup_Size'), r.get('Grid_Size'))
print('%-92s %7s %9s %8s %6s %6s %6s %8s %10s' % ('kernel', 'calls', 'avg us', 'LDS B', 'VGPR', 'AGPR', 'SGPR', 'wg size', 'grid'))
for k, (n, tot, fp) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print('%-92s %7d %9.1f %8s %6s %6s %6s %8s %10s' % (k, n, tot / n, *fp))
