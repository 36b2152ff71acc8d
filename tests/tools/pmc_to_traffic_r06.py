"""Round 5 (kernel names carry the riders flag; read bytes against the algorithmic READ bytes per family: VERDICT r04 item 7): per-launch memory-side traffic of the x3 GEMM families of the default (fp32x3) step from the two rocprofv3 --pmc
passes (FETCH_SIZE, WRITE_SIZE; tests/tools/run_profile_r05.sh), corrected as MI355X_MICROARCH.md (HBM section) prescribes for
gfx950: FETCH_SIZE counts 128-B read requests as 64 B -> x2; WRITE_SIZE is exact; both are reported in KiB.  The Adam kernel (a
pure 16-B/lane stream of known size) is the calibration row.

usage: python tests/tools/pmc_to_traffic_r05.py <dir with pmc_fetch.csv pmc_write.csv> <out.json>
"""
import csv, json, os, re, sys

M, H, I = 2624, 768, 3072


def rows(path):
    return list(csv.DictReader(open(path)))


def pick_all(rs, pred, counter):
    out = []
    for r in rs:
        if r['Counter_Name'] == counter and pred(r['Kernel_Name']):
            out.append((float(r['MeanPerLaunch']), int(r['Launches']), r['Kernel_Name']))
    return out


def x3(akm, bkm, epi):
    # gemm_s3p_kernel<BM, BN, WM, WN, AKM, BKM, ST, KT, NWL, M16, EPI, XTR, SK>
    pat = re.compile(r'gemm_s3p_kernel<\d+, \d+, \d+, \d+, %s, %s, \d+, \d+, \d+, (?:true|false), %d, (?:true|false)(?:, (?:true|false))?>' % (akm, bkm, epi))
    return lambda name: bool(pat.search(name))


# operand bytes a launch must read at least once (x3 operands 6 B / element; aux operands 4 B)
ALG_READ = {
    'gemm_ffn_up_fwd': M * H * 6 + I * H * 6,
    'gemm_dgrad_mul': M * H * 6 + I * H * 6 + M * I * 4,
    'gemm_wgrad': (M * (I + H) * 6 * 2 + M * (3 * H + H) * 6 + M * 2 * H * 6),
}


def main(src, out):
    f, w = rows(os.path.join(src, 'pmc_fetch.csv')), rows(os.path.join(src, 'pmc_write.csv'))
    build = open(os.path.join(src, 'build_info.txt')).read().strip() if os.path.exists(os.path.join(src, 'build_info.txt')) else None
    fams = {
        # family: (predicate, algorithmic bytes per launch (x3 operands 6 B / element, fp32 4 B), launches per layer)
        'gemm_ffn_up_fwd': (x3('false', 'false', 5), M * H * 6 + I * H * 6 + M * I * (6 + 4)),
        'gemm_dgrad_mul': (x3('false', 'true', 6), M * H * 6 + I * H * 6 + M * I * (4 + 6)),
        'gemm_dgrad_add': (x3('false', 'true', 4), None),      # FFN-up and QKV input gradients (average of the two shapes)
        'gemm_wgrad': (lambda n: x3('true', 'true', 0)(n) or x3('true', 'true', 4)(n),
                       (M * (I + H) * 6 * 2 + M * (3 * H + H) * 6 + M * 2 * H * 6) + (2 * I * H + 4 * H * H) * 4),
        'adam': (lambda n: 'adam_kernel' in n, None),
    }
    any_dgrad = re.compile(r'gemm_s3p_kernel<\d+, \d+, \d+, \d+, false, true, ')
    # all four input-gradient products of a layer (FFN-down x gelu', FFN-up + residual (two slabs), attention output (two slabs), QKV + residual (two slabs))
    fams['gemm_dgrad'] = (lambda n: bool(any_dgrad.search(n)),
                          ((M * H * 6 + I * H * 6 + M * I * 10) + (M * I * 6 + I * H * 6 + M * H * 12) + (M * H * 6 + H * H * 6 + M * H * 8)
                           + (M * 3 * H * 6 + 3 * H * H * 6 + M * H * 12)) // 4)
    fams['gemm_dgrad_add'] = (fams['gemm_dgrad_add'][0], ((M * I * 6 + I * H * 6 + M * H * 8) + (M * 3 * H * 6 + 3 * H * H * 6 + M * H * 8)) // 2)
    res = {}
    for name, (pred, alg) in fams.items():
        fr, wr = pick_all(f, pred, 'FETCH_SIZE'), pick_all(w, pred, 'WRITE_SIZE')
        if not fr or not wr:
            continue
        nf = sum(n for _, n, _ in fr)
        fk = sum(v * n for v, n, _ in fr) / nf
        nw = sum(n for _, n, _ in wr)
        wk = sum(v * n for v, n, _ in wr) / nw
        alg_rd = ALG_READ.get(name)
        res[name] = {'launches': nf, 'fetch_size_kib': fk, 'write_size_kib': wk, 'read_bytes': fk * 1024 * 2,
                     'algorithmic_read_bytes': alg_rd, 'read_over_algorithmic': round(fk * 1024 * 2 / alg_rd, 2) if alg_rd else None,
                     'write_bytes': wk * 1024, 'traffic_bytes': fk * 1024 * 2 + wk * 1024, 'algorithmic_bytes': alg,
                     'kernel': fr[0][2][:160], 'build': build,
                     'shape': {'M': M, 'N': I, 'K': H} if name == 'gemm_ffn_up_fwd' else None}
    json.dump(res, open(out, 'w'), indent=1)
    for k, v in res.items():
        print('%-18s %6d launches  read %8.1f MB  written %8.1f MB  algorithmic %s MB  read / algorithmic read %s' % (
            k, v['launches'], v['read_bytes'] / 1e6, v['write_bytes'] / 1e6,
            '%.1f' % (v['algorithmic_bytes'] / 1e6) if v['algorithmic_bytes'] else '-',
            ('%.2f x' % v['read_over_algorithmic']) if v.get('read_over_algorithmic') else '-'))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
