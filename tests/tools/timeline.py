"""Timeline of ONE training step from a rocprofv3 --kernel-trace CSV: per kernel start (relative), duration, queue, and
the idle gap in front of it on its queue; totals per queue.  Usage: timeline.py <kernel_trace.csv> [step_index_from_end]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    return n[:n.index('(')][:60] if '(' in n else n[:60]
ks = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']), r['Queue_Id']) for r in rows))
# a step starts at txt_embed_fwd_kernel
starts = [i for i, k in enumerate(ks) if k[2].startswith('txt_embed_fwd_kernel')]
a, b = starts[-back - 1], starts[-back]
step = ks[a:b]
t0 = step[0][0]
print('step wall %.1f us, %d kernels' % ((ks[b][0] - t0) / 1e3, len(step)))
lastq = {}
busy = collections.Counter()
for s, e, n, q in step:
    gap = (s - lastq[q]) / 1e3 if q in lastq else 0.0
    lastq[q] = max(e, lastq.get(q, 0))
    busy[q] += e - s
    print('%9.1f %7.1f gap %6.1f q%s %s' % ((s - t0) / 1e3, (e - s) / 1e3, gap, q, n))
for q, v in busy.items():
    print('queue', q, 'busy %.1f us' % (v / 1e3))
# union of busy intervals = time with at least one kernel running
iv = sorted((s, e) for s, e, _, _ in step)
cur_s, cur_e, tot = iv[0][0], iv[0][1], 0
for s, e in iv[1:]:
    if s > cur_e: tot += cur_e - cur_s; cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
tot += cur_e - cur_s
print('some kernel running: %.1f us; nothing running: %.1f us' % (tot / 1e3, (ks[b][0] - t0 - tot) / 1e3))
