"""Per-kernel evidence table from the committed rocprofv3 outputs of one bench command: duration (kernel stats), memory-side bytes
per launch (FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE, both in KiB in the PMC summaries) -> GB/s against 8 TB/s,
and matrix-pipe busy (SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x duration x 2.4 GHz)).
Round 5: + matrix-pipe busy over the launch's own active time (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8): the
counter pass serialises kernels, so this is the kernel ALONE at the clock it held) and SQ_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8).
Usage: python tests/tools/pmc_table.py profiles/r02 [bf16|""] [stats prefix: bench (default) | pmc_pass] > profiles/r02_kernel_table[_bf16].md"""
import csv, sys, re
pre = sys.argv[1]; suf = ('_' + sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2] else ''
stats_pre = sys.argv[3] if len(sys.argv) > 3 else 'bench'
if not pre.endswith('/'):
    pre += '_'          # profiles/r02 -> profiles/r02_pmc_fetch.csv; a directory (gpurun_out/r02/) -> .../pmc_fetch.csv
def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    n = n[:n.index('(')] if '(' in n else n
    return n[:110]
def pmc(path, counter):
    out = {}
    try:
        for r in csv.DictReader(open(path)):
            if r['Counter_Name'] == counter:
                out[short(r['Kernel_Name'])] = (int(r['Launches']), float(r['MeanPerLaunch']))
    except OSError:
        pass
    return out
fetch = pmc('%spmc_fetch%s.csv' % (pre, suf), 'FETCH_SIZE')
write = pmc('%spmc_write%s.csv' % (pre, suf), 'WRITE_SIZE')
mfma = pmc('%spmc_mfma%s.csv' % (pre, suf), 'SQ_VALU_MFMA_BUSY_CYCLES')
gui = pmc('%spmc_mfma%s.csv' % (pre, suf), 'GRBM_GUI_ACTIVE')
sqb = pmc('%spmc_mfma%s.csv' % (pre, suf), 'SQ_BUSY_CYCLES')
stats = {}
tot = 0.0
for r in csv.DictReader(open('%s%s%s_kernel_stats.csv' % (pre, stats_pre, suf))):
    k = short(r['Name'])
    stats[k] = (int(r['Calls']), float(r['AverageNs']), float(r['Percentage']))
print('| kernel | launches | avg µs | % of kernel time | read MB / launch | written MB / launch | GB/s | of 8 TB/s | MFMA busy (in-step duration, 2.4 GHz) | MFMA busy (alone, own clock) | SQ busy / active |')
print('|---|---|---|---|---|---|---|---|---|---|---|')
# the counter passes and the timing pass are different runs of the same command with different step counts: per-launch
# averages are comparable only for kernels whose launches scale with the steps alike (not the optimizer, whose 12 whole-buffer
# launches of the "optimizer alone" leg weigh differently)
ratios = sorted(stats[k][0] / fetch[k][0] for k in stats if k in fetch and stats[k][2] > 1.0)
ref_ratio = ratios[len(ratios) // 2] if ratios else None
for k, (calls, avg, pct) in sorted(stats.items(), key=lambda kv: -kv[1][2]):
    if pct < 0.3:
        continue
    f = fetch.get(k); w = write.get(k); mm = mfma.get(k)
    # (5 %: a kernel that skips the first step -- the accumulating form of the weight-gradient launch -- is a few launches short in
    # a 4-step counter pass; the optimizer is excluded by name: the counter passes have no "optimizer alone" leg)
    if f and ref_ratio and (abs(calls / f[0] / ref_ratio - 1) > 0.05 or k.startswith('adam_kernel')):
        f = w = mm = None
    rd = 2 * f[1] * 1024 / 1e6 if f else None           # KiB -> MB, x2: gfx950 FETCH_SIZE tallies 128-B requests at 64 B
    wr = w[1] * 1024 / 1e6 if w else None
    gbs = (rd + wr) * 1e6 / (avg * 1e-9) / 1e9 if (rd is not None and wr is not None) else None
    busy = mm[1] / (1024 * avg * 1e-9 * 2.4e9) if mm else None
    gg = gui.get(k); sb = sqb.get(k)
    busy_own = mm[1] / (1024 * gg[1] / 8) if (mm and gg and gg[1] > 0) else None
    sq_ratio = sb[1] / (gg[1] / 8) if (sb and gg and gg[1] > 0) else None
    print('| `%s` | %d | %.1f | %.1f | %s | %s | %s | %s | %s | %s | %s |' % (
        k, calls, avg / 1e3, pct, '%.1f' % rd if rd is not None else '', '%.1f' % wr if wr is not None else '',
        '%.0f' % gbs if gbs is not None else '', '%.2f' % (gbs / 8000) if gbs is not None else '', '%.2f' % busy if busy is not None else '',
        '%.2f' % busy_own if busy_own is not None else '', '%.1f' % sq_ratio if sq_ratio is not None else ''))
