#!/usr/bin/env python
"""Headline benchmark: train samples/sec of UNITER-base fine-tuning
(BASELINE.json: batch 16 per GPU, 36 regions x 2048-d, 128 text tokens,
12 layers, fp32) on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = forward + BCE loss + backward (+ RCCL gradient all-reduce for N>1)
+ global-norm clip + Adam + LR schedule, dropout ON (p=0.1), synthetic inputs
resident in HBM before the timed region.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time

os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')      # before the HIP runtime initialises: see meme_challenge_amd/__init__.py

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

BASE = dict(attention_probs_dropout_prob=0.1, hidden_act='gelu', hidden_dropout_prob=0.1,
            hidden_size=768, initializer_range=0.02, intermediate_size=3072,
            max_position_embeddings=512, num_attention_heads=12, num_hidden_layers=12,
            type_vocab_size=2, vocab_size=28996)       # == reference config/uniter-base.json
LARGE = dict(BASE, hidden_size=1024, intermediate_size=4096, num_attention_heads=16, num_hidden_layers=24)

# MI355X_MICROARCH.md: fp32 matrix pipe (v_mfma_f32_32x32x2_f32) 157.3; bf16 dense 2500.  'f32x3': the fp32 mode's dense
# products as SIX bf16 MFMA products per block (three bf16 pieces per fp32 value, fp32 accumulate: csrc/gemm_split3.hip) --
# 2500 / 6 = 416.7 TFLOP/s of fp32-equivalent work at the bf16 pipe's peak
PEAK_TFLOPS = {'f32': 157.3, 'f32x3': 2500.0 / 6.0, 'bf16': 2500.0}


def flops_per_step(cfg, B, T, R, L=None, lens=None):
    """Algorithmic FLOPs: 2MNK per GEMM, bwd = 2x fwd, attention
    QK^T and PV included, elementwise excluded.  Returns (total, ffn_only, ffn_up_fwd)."""
    H, I, nl = cfg['hidden_size'], cfg['intermediate_size'], cfg['num_hidden_layers']
    L = T + R if L is None else L         # compacted joint length (max over the batch of tl + nbb)
    M = B * L
    sq = B * L * L
    if lens is not None:                  # token packing: only the valid positions are computed
        M = sum(lens)
        sq = sum(n * n for n in lens)
    per_layer = 2 * M * H * 3 * H + 2 * M * H * H + 2 * 2 * M * H * I + 2 * 2 * sq * H
    fwd = nl * per_layer + 2 * B * R * 2048 * H + 2 * B * H * H + 2 * B * H
    ffn_up_fwd = 2 * M * H * I
    return 3 * fwd, 3 * nl * 2 * ffn_up_fwd, ffn_up_fwd


INTAKE_GBS_PER_CU = 70.0      # MI355X_MICROARCH.md: L2 -> CU ceiling of any 16-byte-per-lane path (LDS-DMA rings: 68-90 GB/s per CU)
HBM_TBS = 8.0


def gemm_floor(M, N, K, BM, BN, KT, pieces, nsplit, cus, hbm_bytes, flops, peak_tflops, staged=True):
    """The bound that applies to ONE launch of a tiled product: the largest of
      mfma_us   = algorithmic FLOPs / the pipe's peak,
      intake_us = bytes staged through LDS / (70 GB/s x CUs the launch can use): tiles x k-tiles x (BM + BN) x KT x 2 B x pieces
                  (pieces = 3 for x3 operands); CUs used = min(CUs, work items = tiles x k-pieces),
      hbm_us    = algorithmic operand + output bytes / 8 TB/s.
    staged=False (the native fp32 kernels stage through registers, not LDS-DMA): no intake term."""
    tiles = -(-M // BM) * -(-N // BN)
    items = tiles * max(1, nsplit)
    nk = -(-K // KT)
    staged_bytes = tiles * nk * (BM + BN) * KT * 2 * pieces if staged else 0
    used = max(1, min(cus, items))
    mfma_us = flops / (peak_tflops * 1e12) * 1e6
    intake_us = staged_bytes / (INTAKE_GBS_PER_CU * 1e9 * used) * 1e6
    hbm_us = hbm_bytes / (HBM_TBS * 1e12) * 1e6
    return {'mfma_us': round(mfma_us, 2), 'intake_us': round(intake_us, 2), 'hbm_us': round(hbm_us, 2),
            'floor_us': round(max(mfma_us, intake_us, hbm_us), 2), 'tiles': tiles, 'items': items, 'tile': '%dx%d' % (BM, BN),
            'staged_mb': round(staged_bytes / 1e6, 1), 'cus_used': used}


def _sum_floors(parts):
    out = {k: round(sum(p[k] for p in parts), 2) for k in ('mfma_us', 'intake_us', 'hbm_us', 'floor_us')}
    out['launches'] = [{k: p[k] for k in ('name', 'tile', 'items', 'cus_used', 'staged_mb', 'mfma_us', 'intake_us', 'hbm_us', 'floor_us')} for p in parts]
    return out


def family_floors(precision, M, H, I, B, L, nh, cus=256, plan=None, sq=None):
    """Per LAYER and family: the floor of each launch (gemm_floor) summed over the family's launches of one layer.
    precision: 'fp32x3' (x3 operands: 6 bytes per value staged, geometry from uniter_gemm_x3_plan), 'bf16' (one piece, 64-deep
    k-tiles) or 'fp32' (native fp32 MFMA kernels: no LDS-DMA staging term).  plan(M, N, K, nsplit_fixed, fwd32) -> (BM, BN, nsplit)
    answers the geometry of a forward / input-gradient product; None = 128 x 128 tiles, the nsplit given."""
    pk = {'fp32x3': 2500.0 / 6.0, 'bf16': 2500.0, 'fp32': 157.3}[precision]
    pieces = 3 if precision == 'fp32x3' else 1
    eb = 2 * pieces if precision != 'fp32' else 4                # bytes per operand value in memory
    KT = 32 if precision == 'fp32x3' else 64
    staged = precision != 'fp32'
    sq = B * L * L if sq is None else sq

    def prod(name, m, n, k, fixed, out_bytes, extra_in=0, fwd32=False):
        BM, BN, ns = plan(m, n, k, fixed, fwd32) if plan is not None else (128, 128, max(1, fixed))
        hb = m * k * eb + n * k * eb + extra_in + (out_bytes if ns == 1 or out_bytes == 0 else m * n * 4 * ns)
        f = gemm_floor(m, n, k, BM, BN, KT, pieces, ns, cus, hb, 2.0 * m * n * k, pk, staged)
        f['name'] = name
        return f

    fam = {}
    # forward (model/layer.py:76-78, :112, :140, :153)
    fam['gemm_qkv_fwd'] = _sum_floors([prod('qkv', M, 3 * H, H, 1, M * 3 * H * (4 if precision != 'bf16' else 2), fwd32=True)])
    fam['gemm_attn_out_fwd'] = _sum_floors([prod('attn_out', M, H, H, 0, M * H * 4)])
    fam['gemm_ffn_up_fwd'] = _sum_floors([prod('ffn_up', M, I, H, 1, M * I * (eb + (4 if precision != 'bf16' else 2)))])
    fam['gemm_ffn_down_fwd'] = _sum_floors([prod('ffn_down', M, H, I, 0, M * H * 4)])
    # input gradients: dU = (g2 W2) * gelu'(u); dy1 = dU W1 + dz2; dctx = g1 Wo; dx = dqkv Wqkv + dz1
    gd = 4 if precision != 'bf16' else 2
    fam['gemm_dgrad'] = _sum_floors([
        prod('d_ffn_down', M, I, H, 1, M * I * eb, extra_in=M * I * gd),
        prod('d_ffn_up', M, H, I, 0, M * H * 4, extra_in=M * H * 4),
        prod('d_attn_out', M, H, H, 0, M * H * 4),
        prod('d_qkv', M, H, 3 * H, 0, M * H * 4, extra_in=M * H * 4)])
    # weight gradients: one grouped launch of whole-K tiles, both operands k-major (fp32x3: 128 x 256 tiles; bf16: 128 x 128)
    wbn = 256 if precision == 'fp32x3' else 128
    shapes = [(I, H), (H, I), (3 * H, H), (H, H)]
    tiles = sum(-(-mo // 128) * -(-no // wbn) for mo, no in shapes)
    nk = -(-M // KT)
    st_bytes = tiles * nk * (128 + wbn) * KT * 2 * pieces if staged else 0
    fl = sum(2.0 * mo * no * M for mo, no in shapes)
    hb = sum(M * (mo + no) * eb + mo * no * 4 for mo, no in shapes)
    used = max(1, min(cus, tiles))
    w = {'name': 'wgrad_group', 'tile': '128x%d' % wbn, 'items': tiles, 'cus_used': used, 'staged_mb': round(st_bytes / 1e6, 1),
         'mfma_us': round(fl / (pk * 1e12) * 1e6, 2), 'intake_us': round(st_bytes / (INTAKE_GBS_PER_CU * 1e9 * used) * 1e6, 2),
         'hbm_us': round(hb / (HBM_TBS * 1e12) * 1e6, 2)}
    w['floor_us'] = max(w['mfma_us'], w['intake_us'], w['hbm_us'])
    fam['gemm_wgrad'] = _sum_floors([w])
    # attention (model/layer.py:85-100): one workgroup per (sample, head); Q, K, V (backward: + dO, O) staged once per workgroup
    pa = pk if precision != 'fp32' else 157.3
    wgs = max(1, min(cus, B * nh))
    qb = 4 if precision != 'bf16' else 2
    a_f = {'name': 'attn_fwd', 'tile': 'head', 'items': B * nh, 'cus_used': wgs, 'staged_mb': round(3 * M * H * qb / 1e6, 1),
           'mfma_us': round(4.0 * sq * H / (pa * 1e12) * 1e6, 2), 'intake_us': round(3 * M * H * qb / (INTAKE_GBS_PER_CU * 1e9 * wgs) * 1e6, 2),
           'hbm_us': round((3 * M * H * qb + M * H * (4 + eb)) / (HBM_TBS * 1e12) * 1e6, 2)}
    a_f['floor_us'] = max(a_f['mfma_us'], a_f['intake_us'], a_f['hbm_us'])
    a_b = {'name': 'attn_bwd', 'tile': 'head', 'items': B * nh, 'cus_used': wgs, 'staged_mb': round((3 * M * H * qb + 2 * M * H * 4) / 1e6, 1),
           'mfma_us': round(8.0 * sq * H / (pa * 1e12) * 1e6, 2),
           'intake_us': round((3 * M * H * qb + 2 * M * H * 4) / (INTAKE_GBS_PER_CU * 1e9 * wgs) * 1e6, 2),
           'hbm_us': round((3 * M * H * qb + 2 * M * H * 4 + 3 * M * H * eb) / (HBM_TBS * 1e12) * 1e6, 2)}
    a_b['floor_us'] = max(a_b['mfma_us'], a_b['intake_us'], a_b['hbm_us'])
    fam['attention_fwd'] = _sum_floors([a_f])
    fam['attention_bwd'] = _sum_floors([a_b])
    # the two dropout + residual + LayerNorm passes of a layer: 16 bytes per element each way (x, residual in; z, y out / dy, z in;
    # dz, dx out) + what the mode adds: the second k-piece slab of the product in front (4 B) and the operand copy of the product
    # behind (x3 pieces 6 B, bf16 2 B)
    extra = {'fp32x3': 4.0 + 6.0, 'bf16': 4.0 + 2.0, 'fp32': 0.0}[precision]
    for name in ('layernorm_fwd', 'layernorm_bwd'):
        us = 2 * (16.0 + extra) * M * H / (HBM_TBS * 1e12) * 1e6
        fam[name] = {'mfma_us': 0.0, 'intake_us': 0.0, 'hbm_us': round(us, 2), 'floor_us': round(us, 2), 'launches': []}
    return fam


def _lscpu_model():
    try:
        import subprocess
        for line in subprocess.run(['lscpu'], capture_output=True, text=True, timeout=10).stdout.splitlines():
            if line.lower().startswith('model name'):
                return line.split(':', 1)[1].strip()
    except Exception:
        pass
    return 'unknown'


def _physical_cores():
    """Physical cores the process may run on: distinct (socket, core) pairs of the CPUs in its affinity mask (lscpu -p);
    falls back to the affinity count."""
    try:
        allowed = os.sched_getaffinity(0)
    except AttributeError:
        allowed = set(range(os.cpu_count() or 1))
    try:
        import subprocess
        out = subprocess.run(['lscpu', '-p=CPU,CORE,SOCKET'], capture_output=True, text=True, timeout=10).stdout
        cores = set()
        for line in out.splitlines():
            if line.startswith('#') or not line.strip():
                continue
            cpu, core, sock = (int(x) if x else 0 for x in line.split(',')[:3])
            if cpu in allowed:
                cores.add((sock, core))
        if cores:
            return len(cores)
    except Exception:
        pass
    return max(1, len(allowed))


def cpu_baseline(seconds_budget=30.0, all_cores=False):
    """Reported baseline only: the CPU oracle (port of the reference forward; autograd backward) on
    BASELINE config 1 (B=4, T=64, R=36, UNITER-base, dropout on), timed on this box's host cores: N = physical cores, 32
    threads where that differs, and one thread (1 warm-up + up to 2 timed steps), all bounded by the budget."""
    import torch
    from oracle import uniter_oracle as O
    from oracle import step_oracle as S
    sd = {k: v.requires_grad_(True) for k, v in O.synth_state_dict(BASE, seed=0).items()}
    b = O.synth_batch(4, 64, 36, seed=1234)
    kw = dict(img_feat=b['img_feat'], img_pos_feat=b['img_pos_feat'], input_ids=b['input_ids'],
              position_ids=b['position_ids'], attention_mask=b['attn_mask'],
              gather_index=b['gather_index'], output_all_encoded_layers=False)
    gf_per_sample = flops_per_step(BASE, 4, 64, 36)[0] / 4 / 1e9          # 52.411

    def step(i):
        drop = O.DropSpec(1234, i, 0.1, 0.1)
        loss = S.bce_with_logits(O.meme_uniter_forward(sd, BASE, drop=drop, **kw), b['labels'], 1.8)
        return torch.autograd.grad(loss, [v for v in sd.values()], allow_unused=True)

    def timed(nthreads, warm, most, budget):
        torch.set_num_threads(nthreads)
        for i in range(warm):
            step(i)
        t0, n = time.perf_counter(), 0
        while n < most and (time.perf_counter() - t0) < budget:
            step(warm + n)
            n += 1
        return (time.perf_counter() - t0) / max(n, 1), n

    hw = os.cpu_count() or 1
    phys = _physical_cores()
    legs = []
    # N = the physical cores the process may use AND the 32-thread figure: the oracle's small CPU ops do not
    # scale past ~32 threads (EPYC 9575F, 128 cores: 0.33 samples/s on 128 threads, 12 s per step, against 6.7 on 32), so the
    # all-cores leg is ONE warm-up-free step when N <= 64 and, beyond that, the figure recorded in round 4 (steps: 0) unless
    # --cpu_all_cores asks for it: 12 s of the driver's time for a number that is noise
    nts = sorted({phys, min(phys, 32)}, reverse=True)
    for nt in nts:
        if nt > 64 and not all_cores and 'EPYC 9575F' not in _lscpu_model():
            #  the recorded figure belongs to ONE CPU model: anywhere else the leg is not run and says so
            legs.append({'cores': nt, 'value': None, 'gflops': None, 'steps': 0, 's_per_step': None,
                         'note': 'not run (about 12 s per step on 128 threads of the pool\'s hosts; --cpu_all_cores runs it); no recorded figure for this CPU model'})
            continue
        if nt > 64 and not all_cores:
            legs.append({'cores': nt, 'value': 0.325, 'gflops': round(0.325 * gf_per_sample, 1), 'steps': 0, 's_per_step': 12.3,
                         'note': 'not run: recorded in round 4 on this CPU model (BENCH_r04.json, one step on 128 threads); --cpu_all_cores runs it'})
            continue
        warm, most = (1, 1) if (nt > 32 and not all_cores) else (2, 8)      # (never time the first call)
        dt, n = timed(nt, warm, most, seconds_budget * 0.3)
        legs.append({'cores': nt, 'value': round(4.0 / dt, 3), 'gflops': round(4.0 / dt * gf_per_sample, 1), 'steps': n, 's_per_step': round(dt, 3)})
    dt1, n1 = timed(1, 1, 2, seconds_budget * 0.3)
    best = max([l for l in legs if l['steps'] > 0], key=lambda l: l['value'])
    torch.set_num_threads(best['cores'])
    return {'value': best['value'], 'unit': 'samples/s', 'cores': best['cores'], 'kind': 'port',
            'gflops': best['gflops'], 'by_cores': legs, 'physical_cores': phys,
            'single_thread': {'value': round(4.0 / dt1, 3), 'gflops': round(4.0 / dt1 * gf_per_sample, 1), 'steps': n1},
            'cpu_model': _lscpu_model(), 'os_cpu_count': hw,
            'sample': 'config 1: UNITER-base B=4 T=64 R=36 fwd+bwd: %s; %d steps after 1 warm-up on 1 '
                      'thread (%.2f s/step); oracle/uniter_oracle.py (torch CPU fp32, dropout on)'
                      % ('; '.join(('%d steps on %d threads (%.3f s/step)' % (l['steps'], l['cores'], l['s_per_step'])) if l['steps'] else
                                   ('%d threads: not run%s' % (l['cores'], (' (recorded %.3f samples/s)' % l['value']) if l['value'] else '')) for l in legs),
                         n1, dt1)}


def pmc_traffic(args, M, cfgd, build_info):
    """Memory-side bytes per launch of the GEMM families from the committed rocprofv3 --pmc passes (separate FETCH_SIZE /
    WRITE_SIZE runs of this command, tests/tools/run_profile_r06.sh -> tests/tools/pmc_to_traffic_r06.py; FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950).  Counters cannot be read from inside a timed run, so this is a PROFILE
    ARTEFACT, reported only for the shape AND the library build it was taken on."""
    here = os.path.dirname(os.path.abspath(__file__))
    if args.workload != 'finetune' or args.precision == 'bf16':
        return None
    if args.precision == 'fp32x3':
        allrec, src = None, None
        for cand in ('profiles/r06_pmc_traffic.json', 'profiles/r05_pmc_traffic.json'):      # the newest set taken on THIS library build
            try:
                rec_ = json.load(open(os.path.join(here, cand)))
            except (OSError, ValueError):
                continue
            b_ = (rec_.get('gemm_ffn_up_fwd') or {}).get('build')
            if allrec is None or (b_ and b_ == build_info):
                allrec, src = rec_, cand
            if b_ and b_ == build_info:
                break
        if allrec is None:
            return None
        rec = allrec.get('gemm_ffn_up_fwd') or {}
        shape = rec.get('shape') or {}
        if (shape.get('M'), shape.get('N'), shape.get('K')) != (M, cfgd['intermediate_size'], cfgd['hidden_size']):
            return None
        if rec.get('build') and rec['build'] != build_info:
            return None
        out = {'source': src}
        for fam in ('gemm_ffn_up_fwd', 'gemm_dgrad', 'gemm_wgrad'):
            r = allrec.get(fam)
            if r:
                out[fam] = {'bytes_per_launch': int(r['traffic_bytes']), 'algorithmic_bytes': int(r['algorithmic_bytes']),
                            'read_bytes': int(r['read_bytes']), 'write_bytes': int(r['write_bytes']), 'kernel': r.get('kernel')}
        return out
    src = 'profiles/r03_pmc_traffic.json'
    try:
        allrec = json.load(open(os.path.join(here, src)))
        rec = allrec['ffn_up_fwd']
    except (OSError, KeyError, ValueError):
        return None
    shape = rec.get('shape', {})
    if (shape.get('M'), shape.get('N'), shape.get('K')) != (M, cfgd['intermediate_size'], cfgd['hidden_size']):
        return None
    if rec.get('build') and rec['build'] != build_info:
        return None
    out = {'source': src, 'gemm_ffn_up_fwd': {'bytes_per_launch': int(rec['traffic_bytes']), 'algorithmic_bytes': int(rec['algorithmic_bytes'])}}
    wg = allrec.get('wgrad') or allrec.get('wgrad_stream_k')
    if wg and wg.get('build') == rec.get('build'):
        # the weight-gradient family of the fp32 step (average over its four shapes)
        out['gemm_wgrad'] = {'bytes_per_launch': int(wg['traffic_bytes']), 'algorithmic_bytes': int(wg['algorithmic_bytes_avg']),
                             'kernel': wg.get('kernel'),
                             'note': 'reads served by L2 misses: the operand set stays in the Infinity Cache (FETCH_SIZE counts L2 fills)'}
    return out


def parse_rccl_log(path):
    """What RCCL says it built, from rank 0's NCCL_DEBUG=INFO log (INIT,GRAPH,TUNING): channel count, ring / tree graphs,
    and the algorithm / protocol of the collectives where the TUNING lines name them.  Every field is None when the log has no
    such line (the format is RCCL's, not a contract)."""
    import re
    out = {'channels': None, 'graphs': None, 'algorithm': None, 'protocol': None, 'log_lines': 0}
    try:
        text = open(path, errors='replace').read()
    except OSError:
        return out
    lines = text.splitlines()
    out['log_lines'] = len(lines)
    ch = [int(m.group(2)) for m in re.finditer(r'Channel (\d+)/(\d+)', text)]
    if ch:
        out['channels'] = max(ch)
    else:
        m = re.search(r'(\d+) coll channels', text) or re.search(r'nChannels (\d+)', text)
        if m:
            out['channels'] = int(m.group(1))
    graphs = sorted({g for g in ('Ring', 'Tree', 'CollNet', 'NVLS') if re.search(r'\b%s\b' % g, text)})
    out['graphs'] = graphs or None
    algos = re.findall(r'[Aa]lgo(?:rithm)?[ =:]+(\w+)', text)
    protos = re.findall(r'[Pp]roto(?:col)?[ =:]+(\w+)', text)
    if algos:
        out['algorithm'] = max(set(algos), key=algos.count)
    if protos:
        out['protocol'] = max(set(protos), key=protos.count)
    return out


def comm_block(sync, steps, rccl_log):
    """The `comm` block of the line (N > 1 or a forced one-rank exchange): per collective of a step its payload bytes, the time
    from its issue to the moment the optimizer's stream had it, and the part of that the optimizer's stream WAITED (exposed);
    payload dtype; and RCCL's own report.  Times are means over the timed steps (dp.GradSync.collect_timings)."""
    sync.collect_timings()
    recs = sync.timings
    by = {}
    for r in recs:
        by.setdefault((r['start'], r['end']), []).append(r)
    cols = []
    for (s0, e0), rs in sorted(by.items()):
        cols.append({'elements': e0 - s0, 'bytes': rs[0]['bytes'], 'per_step': round(len(rs) / max(steps, 1), 2),
                     'issue_to_done_ms': round(sum(r['issue_to_done_ms'] for r in rs) / len(rs), 4),
                     'exposed_ms': round(sum(r['exposed_ms'] for r in rs) / len(rs), 4)})
    out = {'payload': sync.payload, 'world': sync.world, 'collectives': cols,
           'bytes_per_step': int(sum(c['bytes'] * c['per_step'] for c in cols)),
           'exposed_ms_per_step': round(sum(c['exposed_ms'] * c['per_step'] for c in cols), 4),
           'sparse_steps': sync.sparse_steps,
           'sparse_rows_per_rank': getattr(sync, 'last_sparse_rows', None),
           'note': 'dense all-reduces only are timed (issue: event on the issuing stream in front of the collective; done: event behind '
                   'the consumer stream\'s wait); the sparse row exchange is summed in place on its issuing stream'}
    # the design's arithmetic beside the measurement: the last dense collective of a step has no backward
    # compute left to hide behind -- a ring all-reduce moves 2 (N - 1) / N of its bytes per rank, at one link's rate or at all seven
    if cols:
        from meme_challenge_amd import dp as _dp
        last = cols[-1]          # (sorted by offset in the flat buffer: the embeddings close it, and they are issued last)
        out['predicted_exposed_ms'] = {'last_collective_bytes': last['bytes'],
                                       'one_link_ring': round(_dp.predicted_exposed_ms(last['bytes'], sync.world, 1), 4),
                                       'all_links': round(_dp.predicted_exposed_ms(last['bytes'], sync.world, 7), 4),
                                       'note': '2 (N - 1) / N x bytes / (153 GB/s x links): arithmetic for the collective issued last (the embeddings\' bucket), not a measurement'}
    if rccl_log:
        out['rccl'] = parse_rccl_log(rccl_log)
    sync.timings = []
    return out


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=16, help='per-GPU batch')
    ap.add_argument('--txt_len', type=int, default=128)
    ap.add_argument('--num_bb', type=int, default=36)
    ap.add_argument('--model', choices=['base', 'large'], default='base')
    ap.add_argument('--precision', choices=['fp32x3', 'fp32', 'bf16'], default='fp32x3',
                    help="fp32x3 (default: BASELINE configs[1] in fp32 arithmetic -- the encoder's dense products as six bf16 MFMA "
                         "products per block on three bf16 pieces per fp32 value, fp32 accumulate, no less accurate than the fp32 "
                         "MFMA kernels), fp32 (the same step on the native fp32 MFMA kernels), or bf16: bf16 MFMA for the dense "
                         "GEMMs, fp32 elsewhere (configs[2])")
    ap.add_argument('--no_native_leg', action='store_true',
                    help='fp32x3 at N = 1: skip the native-fp32 timing of the same step in the same process (native_fp32 in the line)')
    ap.add_argument('--workload', choices=['finetune', 'multitask'], default='finetune',
                    help='finetune = MemeUniter step (BASELINE configs[1-3], default); multitask = UNITER + ITM/MLM/MRFR '
                         'heads, one task drawn per step (configs[4]; use --batch 32)')
    ap.add_argument('--ragged', action='store_true',
                    help='per-sample lengths tl ~ U{8..T}, nbb ~ U{10..R} (SURVEY 8(d) D1: correctness/reporting variant, not the headline)')
    ap.add_argument('--packed', action='store_true',
                    help='token packing: compute the valid positions only (pays off with --ragged; identical results)')
    ap.add_argument('--no_cpu_baseline', action='store_true')
    ap.add_argument('--cpu_all_cores', action='store_true', help='CPU baseline: run the all-physical-cores leg in full (2 warm-ups + up to 8 steps)')
    ap.add_argument('--no_bf16_leg', action='store_true',
                    help='fp32x3 at N = 1: skip the bf16-mode timing of the same step in the same process (bf16 in the line: BASELINE configs[2] arithmetic)')
    ap.add_argument('--no_side_stream', action='store_true')
    ap.add_argument('--prewarm_s', type=float, default=1.0, help='seconds of untimed steps before the W warm-up steps (clock ramp)')
    ap.add_argument('--no_adam_overlap', action='store_true',
                    help='run the optimizer step as one launch on the main stream instead of block by block beside the next forward')
    ap.add_argument('--prof_kind', type=int, default=-1, help='UNITER_K_* kind timed with HIP events inside the timed region (-1 = every kind, 0 = none)')
    ap.add_argument('--reserve_ab', action='store_true',
                    help='with a gradient exchange attached: time the step once more with no CUs reserved for its kernels '
                         '(comm.no_reserve; `value` is the default reserve, comm.cu_reserve)')
    ap.add_argument('--no_reserve_pick', action='store_true',
                    help='N > 1: keep the attach-time CU reserve (dp.cu_reserve_default) instead of timing a few untimed steps with a reserve '
                         'of 16, 0 and 48 CUs before the warm-up and keeping the fastest (comm.reserve_pick)')
    ap.add_argument('--reserve_pick_budget_s', type=float, default=20.0,
                    help='N > 1: wall-time budget of the CU-reserve measurement in front of the warm-up (candidates that do not fit are '
                         'skipped on every rank alike; comm.reserve_pick.wall_s says what it took)')
    ap.add_argument('--both_exchanges', action='store_true',
                    help='N > 1: time the other form of the word-embedding exchange too (a second timed region, listed in '
                         'comm.other_exchange; `value` is always the requested exchange)')
    ap.add_argument('--dp_sparse_embeddings', action='store_true',
                    help='N > 1: exchange the touched word-embedding gradient rows (all-gather of ids + rows) instead of '
                         'all-reducing the whole 28996 x 768 table')
    return ap.parse_args(argv)


def rank_host_threads(world):
    """CPU threads a rank's host-side pools get: the cores it may use divided by the ranks of the node, 8 at most, 1 at least."""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    return max(1, min(8, cores // max(1, world)))


def launch_ranks(n, cmd, env=None, poll_s=0.2):
    """`python bench.py --gpus N` outside a launcher: THIS process -- which has made no GPU call and makes none -- starts N
    fresh rank processes (one per device; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as torch.distributed.run sets them),
    relays rank 0's stdout (the JSON line), sends the other ranks' stdout to stderr, and returns 0 only if every rank
    did.  A rank that fails takes the others down with it (exact PIDs, no pattern kill) instead of leaving them in a
    collective.  Nothing is re-executed: the parent stays a plain supervisor."""
    import socket
    with socket.socket() as s:                       # a free rendezvous port on the loopback interface
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    base = dict(os.environ if env is None else env)
    base.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')          # dmabuf IPC: the only form this pool's driver supports
    # host threads per rank: N ranks x an uncapped OpenMP / torch pool each (256 hardware threads on the pool's hosts) cost the small
    # host-side ops 16 x (the CLI trainer's own finding); the caller's setting stands
    per_rank = str(rank_host_threads(n))
    for var in ('OMP_NUM_THREADS', 'MKL_NUM_THREADS'):
        base.setdefault(var, per_rank)
    base['UNITER_BENCH_LAUNCHER'] = 'self'
    procs = []
    for r in range(n):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                 MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        procs.append(subprocess.Popen(list(cmd), env=e, stdout=None if r == 0 else sys.stderr))
    rc = 0
    try:
        live = list(procs)
        while live and rc == 0:
            time.sleep(poll_s)
            for p in list(live):
                code = p.poll()
                if code is None:
                    continue
                live.remove(p)
                if code != 0:
                    rc = code if code > 0 else 1
                    print('bench.py: rank %d exited with code %d; stopping the other ranks' % (procs.index(p), code),
                          file=sys.stderr)
                    break
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=30)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    return rc


def main(argv=None):
    args = parse_args(argv)
    if args.gpus > 1 and 'RANK' not in os.environ:
        # no launcher environment: be the launcher (N = 1 stays in this process: the driver's BENCH line is unchanged)
        return launch_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + list(sys.argv[1:] if argv is None else argv))
    return run_rank(args)


def run_rank(args):
    # ONE line on stdout: libraries under us write to file descriptor 1 as well (gloo's connection notes, RCCL's version banner
    # under NCCL_DEBUG=INFO).  Descriptor 1 points at stderr for the run; the JSON line goes to the saved descriptor, which
    # becomes descriptor 1 again when the run ends (an in-process caller of main() keeps its stdout).
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    state = {'rccl_log': None}
    try:
        return _run_rank(args, real_stdout, state)
    finally:
        sys.stdout.flush()
        if not _AS_SCRIPT:
            # an in-process caller keeps its stdout.  As a script the process ends here and descriptor 1 STAYS on stderr: RCCL
            # prints its version banner through C stdio when the process exits, and the line on stdout must stay one line
            os.dup2(real_stdout, 1)
            os.close(real_stdout)
        if state['rccl_log']:
            try:
                os.remove(state['rccl_log'])
            except OSError:
                pass


def _run_rank(args, real_stdout, state):
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        if rank == 0:
            print('warning: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE' % (args.gpus, world), file=sys.stderr)
    # RCCL ('nccl'); UNITER_DIST_BACKEND=gloo lets several ranks share one GPU (tests: RCCL wants one device per rank)
    backend = os.environ.get('UNITER_DIST_BACKEND', 'nccl')
    ndev = torch.cuda.device_count()                  # does not initialise the GPU
    if ndev == 0:
        raise SystemExit('bench.py needs an MI355X: no GPU visible (there is no CPU path)')
    if local_rank >= ndev and backend == 'nccl':
        raise SystemExit('bench.py: rank %d has no device (%d visible); RCCL needs one GPU per rank' % (local_rank, ndev))
    local_dev = local_rank % ndev
    torch.cuda.set_device(local_dev)
    dev = torch.device('cuda', local_dev)
    use_dist = world > 1 or 'RANK' in os.environ
    if world > 1:
        # (a rank started by torch.distributed.run has no launcher of ours that set OMP_NUM_THREADS: cap the pools here as well)
        torch.set_num_threads(int(os.environ.get('OMP_NUM_THREADS') or rank_host_threads(world)))
    rccl_log = None
    if use_dist and rank == 0 and backend == 'nccl' and 'NCCL_DEBUG' not in os.environ:
        # RCCL's own account of what it built (rings / trees, channels) for the `comm` block of the line: rank 0's INFO log
        # goes to a file (NCCL_DEBUG_FILE), parsed after the run
        import tempfile
        rccl_log = state['rccl_log'] = os.path.join(tempfile.gettempdir(), 'uniter_rccl_rank0_%d.log' % os.getpid())
        os.environ.update(NCCL_DEBUG='INFO', NCCL_DEBUG_SUBSYS='INIT,GRAPH,TUNING', NCCL_DEBUG_FILE=rccl_log)
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        from meme_challenge_amd import dp as _dp0
        _dp0.prepare_rccl_env(world)         # (caps RCCL's channels at the CU reserve only under UNITER_DP_CAP_CHANNELS=1)
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from meme_challenge_amd import _lib
    from meme_challenge_amd.model import UniterConfig, UniterModel
    from meme_challenge_amd.meme_uniter import MemeUniter
    from meme_challenge_amd.trainer import FusedAdam, TrainStep, get_scheduler, sync_step
    from meme_challenge_amd.utils import make_synthetic_batch, make_synthetic_pretrain_batch
    from meme_challenge_amd import dp
    import ctypes as C

    cfgd = BASE if args.model == 'base' else LARGE
    torch.manual_seed(0)                                   # identical init on every rank
    cfg = UniterConfig.from_dict(cfgd)
    B, T, R = args.batch, args.txt_len, args.num_bb
    lens = {}
    if args.ragged:
        import numpy as np
        rng = np.random.Generator(np.random.PCG64(4321 + rank))
        lens = dict(txt_lens=[int(x) for x in rng.integers(8, T + 1, size=B)],
                    num_bbs=[int(x) for x in rng.integers(10, R + 1, size=B)])
    config = dict(optimizer='adam', lr=3e-5, beta1=0.9, beta2=0.999, weight_decay=1e-3,
                  gradient_accumulation=1, max_grad_norm=5, pos_wt=1.8, loss_func='bce_logits',
                  scheduler='warmup_cosine', warmup_steps=500, max_epoch=30)
    if args.workload == 'finetune':
        model = MemeUniter(UniterModel(cfg, img_dim=2048), cfg.hidden_size, 1).to(dev).train()
        encoder = model.uniter_model
        batch = make_synthetic_batch(B, T, R, seed=1234 + rank, device=dev, **lens)
    else:
        from meme_challenge_amd.pretrain import UniterForPretraining
        model = UniterForPretraining(cfg, img_dim=2048, img_label_dim=1601).to(dev).train()
        encoder = model.uniter
        tasks = ('itm', 'mlm', 'mrfr')
        batches = {t: make_synthetic_pretrain_batch(t, B, T, R, seed=1234 + rank, device=dev, **lens) for t in tasks}
        import random
        task_rng = random.Random(99)                       # same task sequence on every rank
    encoder.use_side_stream = not args.no_side_stream
    encoder.precision = args.precision
    encoder.pack_padded = args.packed
    encoder.set_dropout_seed(1234 + rank, 0)
    opt = FusedAdam(model, lr=config['lr'], weight_decay=config['weight_decay'])
    if not args.no_adam_overlap:
        opt.overlap_encoder = encoder
    sched = get_scheduler(opt, config, steps_per_epoch=1000)
    sync = None
    if use_dist:
        dp.broadcast_parameters(model)
        sync = dp.attach(model, sparse_embeddings=args.dp_sparse_embeddings, accum=config['gradient_accumulation'], token_capacity=B * T)
        sync.timing = True
    if args.workload == 'finetune':
        step = TrainStep(model, opt, sched, config, grad_sync=sync)

        def one_step():
            step.train_iter(batch, iters=0)
    else:
        last = {}
        if sync is None:
            opt.attach_norm_hooks(encoder)
        opt.lazy_zero_encoder = encoder

        def one_step():
            task = task_rng.choice(tasks)
            if sync is not None:     # the MLM task's tied decoder makes the word-embedding gradient dense
                sync.prepare(True, token_ids=None if task == 'mlm' else batches[task]['input_ids'])
            loss = model(batches[task], task, compute_loss=True).mean()
            loss.backward()
            sync_step(opt, sync, 1, config['max_grad_norm'])
            sched.step()
            last['loss'] = loss.detach()

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # clock warm-up (untimed, not part of W): a GPU that idled through the imports starts at its lowest clocks; about a
    # second of steps before the W warm-up steps keeps the ramp out of the timed region
    t_pre = time.perf_counter()
    while args.prewarm_s > 0 and time.perf_counter() - t_pre < args.prewarm_s:
        for _ in range(5):
            one_step()
        torch.cuda.synchronize()
    # N > 1: how many CUs the persistent matrix kernels should leave to RCCL is a property of the node (channels RCCL opens on its
    # links, how long its kernels run beside the backward pass): time a few untimed steps with a reserve of 16, 0 and 48 CUs,
    # keep the fastest on every rank (dp.pick_cu_reserve; UNITER_DP_CU_RESERVE or --no_reserve_pick fix it instead)
    reserve_pick = None
    if sync is not None and world > 1 and not args.no_reserve_pick and 'UNITER_DP_CU_RESERVE' not in os.environ:
        reserve_pick = dp.pick_cu_reserve(sync, encoder, one_step, steps=max(1, min(6, args.steps)), warm=max(1, min(2, args.warmup)),
                                          budget_s=args.reserve_pick_budget_s)
    for _ in range(args.warmup):
        one_step()
    lib = _lib.lib()
    handle = encoder._handle
    barrier()
    if args.workload == 'multitask':
        task_rng.seed(99)            # the timed steps draw the same task sequence whatever ran before them
    if args.prof_kind:
        # in-kernel launch stamps of every GEMM (two atomics per wave, nothing added to the streams)
        _lib.check(lib.uniter_prof_enable_stamps(handle, 1, None))
    if sync is not None:
        sync.collect_timings()           # drop the warm-up's records
        sync.timings = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    barrier()
    dt = time.perf_counter() - t0
    comm = None
    if sync is not None:
        comm = comm_block(sync, args.steps, rccl_log if rank == 0 else None)
        comm['sparse_embeddings'] = bool(args.dp_sparse_embeddings)
        # CUs the persistent matrix kernels leave to RCCL's kernels (dp.cu_reserve_default: UNITER_DP_CU_RESERVE, else 16 with more
        # than one rank) and the channel cap RCCL was started with
        comm['cu_reserve'] = int(getattr(sync, 'cu_reserve', 0))
        comm['rccl_max_nchannels_env'] = os.environ.get('NCCL_MAX_NCHANNELS')
        if reserve_pick is not None:
            comm['reserve_pick'] = reserve_pick
        # --both_exchanges (N > 1, finetune): the same timed region once more with the OTHER form of the word-embedding exchange
        # (dense table all-reduce <-> touched rows only).  `value` stays the REQUESTED exchange's; the other one is listed
        # beside it in comm.other_exchange.  Auxiliary: every rank runs it under try / except and the ranks agree on an ok
        # flag before anyone uses a figure of it -- a failure leaves the already-measured line untouched
        if args.reserve_ab and args.workload == 'finetune':
            # --reserve_ab: the same timed region with NO CUs reserved (every persistent launch on all 256), listed in
            # comm.no_reserve; `value` stays the default's.  Guarded like --both_exchanges
            ok, alt, err, dt3 = 1, None, None, 0.0
            try:
                encoder.cu_reserve = 0
                for _ in range(max(3, args.warmup // 2)):
                    one_step()
                barrier()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    one_step()
                barrier()
                dt3 = time.perf_counter() - t1
            except Exception as e:                                   # noqa: BLE001
                ok, err = 0, repr(e)
            try:
                flag = torch.tensor([float(ok), dt3], dtype=torch.float64, device=dev)
                if use_dist:
                    dist.all_reduce(flag[:1], op=dist.ReduceOp.MIN)
                    dist.all_reduce(flag[1:], op=dist.ReduceOp.MAX)
                ok, dt3 = int(flag[0].item()), float(flag[1].item())
                if ok:
                    alt = {'cu_reserve': 0, 'ms_per_step': round(dt3 / args.steps * 1e3, 3), 'value': round(B * world * args.steps / dt3, 2)}
            except Exception as e:                                   # noqa: BLE001
                alt, err = None, err or repr(e)
            comm['no_reserve'] = alt if alt is not None else {'error': err or 'failed on another rank'}
            encoder.cu_reserve = int(getattr(sync, 'cu_reserve', 0))
            sync.collect_timings()
            sync.timings = []
        if args.both_exchanges and world > 1 and args.workload == 'finetune':
            ok, alt, err = 1, None, None
            alt_sparse = not args.dp_sparse_embeddings
            try:
                sync2 = dp.attach(model, sparse_embeddings=alt_sparse, accum=config['gradient_accumulation'])
                sync2.timing = True
                step.grad_sync = sync2
                for _ in range(max(3, args.warmup // 2)):
                    one_step()
                barrier()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    one_step()
                barrier()
                dt2 = time.perf_counter() - t1
            except Exception as e:                                   # noqa: BLE001
                ok, err = 0, repr(e)
            try:
                flag = torch.tensor([float(ok), dt2 if ok else 0.0], dtype=torch.float64, device=dev)
                dist.all_reduce(flag[:1], op=dist.ReduceOp.MIN)
                dist.all_reduce(flag[1:], op=dist.ReduceOp.MAX)
                ok, dt2 = int(flag[0].item()), float(flag[1].item())
                if ok:
                    alt = comm_block(sync2, args.steps, None)
                    alt['sparse_embeddings'] = alt_sparse
                    alt['ms_per_step'] = round(dt2 / args.steps * 1e3, 3)
                    alt['value'] = round(B * world * args.steps / dt2, 2)
            except Exception as e:                                   # noqa: BLE001
                alt, err = None, err or repr(e)
            comm['other_exchange'] = alt if alt is not None else {'error': err or 'failed on another rank'}
            step.grad_sync = sync
            um = getattr(model, 'uniter_model', None)
            if um is not None:
                um._grad_hook = sync.hook
    NK = 11
    k_n, k_ms = (C.c_int * NK)(), (C.c_double * NK)()          # GEMM families: stamps taken INSIDE the timed region
    e_n, e_ms = (C.c_int * NK)(), (C.c_double * NK)()          # attention / LayerNorm: HIP events in a separate pass
    EV_STEPS = 5
    prof_error = None
    bwd_union_ms = C.c_double(0.0)       # time with an input- or weight-gradient GEMM running (the two share the chip)
    if args.prof_kind:
        try:      # per-family figures are auxiliary: a failure here is reported in the line, it does not lose the line
            _lib.check(lib.uniter_prof_collect_stamps(handle, k_n, k_ms, NK))
            _lib.check(lib.uniter_prof_stamps_union(handle, (1 << 6) | (1 << 7), C.byref(bwd_union_ms)))
            _lib.check(lib.uniter_prof_enable_stamps(handle, 0, None))
            # event pairs around every launch cost ~7 us each and serialise the two backward streams (fp32 step +11 %,
            # bf16 +27 %): they stay out of the timed region; this pass only times the kernels that carry no stamps
            _lib.check(lib.uniter_prof_enable(handle, -1))
            te = time.perf_counter()
            for _ in range(EV_STEPS):
                one_step()
            torch.cuda.synchronize()
            ev_ms = (time.perf_counter() - te) / EV_STEPS * 1e3
            _lib.check(lib.uniter_prof_collect_kinds(handle, e_n, e_ms, NK))
            _lib.check(lib.uniter_prof_enable(handle, 0))
        except Exception as e:                                   # noqa: BLE001
            prof_error = repr(e)
            ev_ms = 0.0
            for k in range(NK):
                k_n[k] = 0; e_n[k] = 0
    loss = float((step.last_loss if args.workload == 'finetune' else last['loss']).item())
    native = None
    if args.precision == 'fp32x3' and world == 1 and not use_dist and not args.no_native_leg:
        # the same step on the native fp32 MFMA kernels, same process and box: what the six-product form is measured against
        try:
            st_ = model.param_store()
            encoder.precision = 'fp32'
            saved_mirror, st_.mirror = getattr(st_, 'mirror', None), None      # (the optimizer would keep writing the pieces)
            for _ in range(5):
                one_step()
            torch.cuda.synchronize()
            nsteps = min(args.steps, 20)
            tn = time.perf_counter()
            for _ in range(nsteps):
                one_step()
            torch.cuda.synchronize()
            dtn = time.perf_counter() - tn
            native = {'value': round(B * nsteps / dtn, 2), 'ms_per_step': round(dtn / nsteps * 1e3, 3), 'steps': nsteps,
                      'kernels': 'gemm_f32_v3_kernel (v_mfma_f32_32x32x2_f32) for every dense product and the fp32-MFMA attention of '
                                 'attention_f32.hip (precision fp32, the library default); everything else identical'}
            st_.mirror = saved_mirror
            st_.mirror_dirty = True
            encoder.precision = 'fp32x3'
            one_step()                                   # back in the timed mode (the weight pieces are refreshed here)
            torch.cuda.synchronize()
        except Exception as e:                               # noqa: BLE001
            native = {'error': repr(e)}
    bf16_leg = None
    if args.precision == 'fp32x3' and world == 1 and not use_dist and not args.no_bf16_leg and args.workload == 'finetune':
        # the same step in the bf16 mode (BASELINE configs[2] arithmetic on one GPU), same process and box, 20 timed steps: the
        # driver's line carries a bf16 figure too.  Auxiliary: a failure is reported in the block, the headline is untouched
        try:
            st_ = model.param_store()
            encoder.precision = 'bf16'
            for _ in range(8):
                one_step()
            torch.cuda.synchronize()
            nsteps = min(args.steps, 20)
            tb = time.perf_counter()
            for _ in range(nsteps):
                one_step()
            torch.cuda.synchronize()
            dtb = time.perf_counter() - tb
            tot_b = flops_per_step(cfgd, B, T, R, int(batch['attn_mask'].shape[1]), batch['seq_lens'] if args.packed else None)[0]
            bf16_leg = {'value': round(B * nsteps / dtb, 2), 'ms_per_step': round(dtb / nsteps * 1e3, 3), 'steps': nsteps,
                        'step_mfma_frac': round(tot_b / (dtb / nsteps) / (PEAK_TFLOPS['bf16'] * 1e12), 4), 'peak_tflops': PEAK_TFLOPS['bf16'],
                        'dtype': 'bf16', 'final_loss': round(float(step.last_loss.item()), 5),
                        'kernels': 'precision bf16: gemm_dma_kernel / gemm_dma_wgrad_group_kernel on bf16-resident operands, '
                                   'attention on the bf16 pipe, fp32 master weights / LayerNorm / loss / optimizer; same batch, same optimizer'}
            encoder.precision = 'fp32x3'
            st_.mirror_dirty = True
            one_step()                                   # back in the timed mode (the weight pieces are rebuilt here)
            torch.cuda.synchronize()
        except Exception as e:                               # noqa: BLE001
            bf16_leg = {'error': repr(e)}
            encoder.precision = 'fp32x3'
            try:      #  the later legs must not run on weight pieces the failed leg left stale
                model.param_store().mirror_dirty = True
            except Exception:                                # noqa: BLE001
                pass
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = B * world * args.steps / dt
        L_eff = int((batch if args.workload == 'finetune' else batches['itm'])
                    ['attn_mask' if args.workload == 'finetune' else 'attn_masks'].shape[1])
        cur = batch if args.workload == 'finetune' else batches['itm']
        total, ffn, ffn_up = flops_per_step(cfgd, B, T, R, L_eff, cur['seq_lens'] if args.packed else None)
        M_eff = sum(cur['seq_lens']) if args.packed else B * L_eff
        dt_name = 'f32' if args.precision in ('fp32', 'fp32x3') else 'bf16'       # the arithmetic type of the path's results
        pk_name = {'fp32': 'f32', 'fp32x3': 'f32x3', 'bf16': 'bf16'}[args.precision]
        peak = PEAK_TFLOPS[pk_name]                    # the dense products' pipe
        # attention: fp32 MFMAs in the fp32 mode; in the fp32x3 mode its products are x3 products too (L <= 192,
        # csrc/attention_x3.hip; UNITER_ATTN_X3=0 keeps the fp32 MFMAs); bf16 mode: the bf16 pipe
        attn_x3 = args.precision == 'fp32x3' and L_eff <= 192 and os.environ.get('UNITER_ATTN_X3', '1')[:1] != '0'
        peak_attn = PEAK_TFLOPS['bf16' if args.precision == 'bf16' else ('f32x3' if attn_x3 else 'f32')]
        H, I, nl = cfgd['hidden_size'], cfgd['intermediate_size'], cfgd['num_hidden_layers']
        sq = sum(n * n for n in cur['seq_lens']) if args.packed else B * L_eff * L_eff
        g_qkv, g_o, g_ffn = 2.0 * M_eff * H * 3 * H, 2.0 * M_eff * H * H, 2.0 * M_eff * H * I
        att = 2.0 * 2 * sq * H
        # algorithmic work per step of each timed family (BASELINE.md section 3 conventions; LayerNorm in bytes:
        # forward reads x, residual and writes z, y = 16 B / element, backward reads dy, z and writes dz, dx = 16 B / element)
        fam = {1: ('gemm_ffn_up_fwd', 'mfma', nl * g_ffn), 2: ('gemm_ffn_down_fwd', 'mfma', nl * g_ffn),
               3: ('gemm_qkv_fwd', 'mfma', nl * g_qkv), 4: ('gemm_attn_out_fwd', 'mfma', nl * g_o),
               5: ('attention_fwd', 'mfma', nl * att), 6: ('gemm_dgrad', 'mfma', nl * (g_qkv + g_o + 2 * g_ffn)),
               7: ('gemm_wgrad', 'mfma', nl * (g_qkv + g_o + 2 * g_ffn)), 8: ('attention_bwd', 'mfma', nl * 2 * att),
               9: ('layernorm_fwd', 'hbm', nl * 2 * 16.0 * M_eff * H), 10: ('layernorm_bwd', 'hbm', nl * 2 * 16.0 * M_eff * H)}
        x3k = ('gemm_s3p_kernel (csrc/gemm_split3.hip: x3 operands, six v_mfma_f32_16x16x32_bf16 per block, loader + compute waves; 128 x 256 tiles '
               'for the wide products and the grouped weight gradients, persistent 128 x 128 tiles elsewhere: uniter_gemm_x3_plan)')
        kernel_of = {'f32x3': {1: x3k + ', bias + GELU epilogue, activation out as x3 pieces + gelu\' fp32', 2: x3k + ', two k-pieces (slabs summed by the LayerNorm pass)',
                               3: x3k, 4: x3k + ', two k-pieces', 6: x3k + ' (weights k-major: ds_read_b64_tr_b16)',
                               7: x3k + ': the four weight gradients of a layer in ONE launch of whole-K tiles (both operands k-major) that also carries the '
                                        'layer\'s bias / LayerNorm column reductions and clip-norm partial sums (riders)'},
                     'f32': {1: 'gemm_f32_v3_kernel<64,64,false,false,TAG=1>', 6: 'gemm_f32_v3_kernel<64,64,false,true,...>',
                             7: 'gemm_f32_v3_kernel<64,64,true,true,0,false> (whole-K 64x64 tiles; UNITER_WGRAD_WHOLE=0: the stream-K form); layer 0: '
                                'gemm_f32_wgrad_group_kernel (its four products as one launch)'},
                     'bf16': {1: 'gemm_dma_kernel<128,128,false,false,SWAP,2,EPI=5> (bias + GELU + gelu\' bf16)',
                              6: 'gemm_dma_kernel<128,128,false,true,SWAP,2,EPI>', 7: 'gemm_dma_wgrad_group_kernel<2> (the four weight gradients of a layer in one launch of whole-K 128x128 tiles)'}}
        # the bound that applies, per family: floor = max(MFMA time at peak, LDS-staged bytes / (70 GB/s x CUs
        # used), algorithmic HBM bytes / 8 TB/s) of every launch, summed over a layer's launches of the family, x layers
        def _plan_x3(m_, n_, k_, fixed, fwd32):
            c_, n2 = C.c_int(0), C.c_int(0)
            fn = lib.uniter_gemm_x3_plan_fwd32 if fwd32 else lib.uniter_gemm_x3_plan
            _lib.check(fn(m_, n_, k_, 0, fixed, C.byref(c_), C.byref(n2)))
            return (128, {3: 128, 4: 256, 5: 192}.get(c_.value, 128), n2.value)

        def _plan_b16(m_, n_, k_, fixed, fwd32):
            # gemm_bf16_dma.hip: 128 x 128 tiles (64 x 128 for the few-tile one-piece products), k-pieces by gemm_bf16v2_pick_split
            tiles, nk = -(-m_ // 128) * -(-n_ // 128), -(-k_ // 64)
            ns = fixed if fixed > 0 else (4 if (tiles <= 128 and nk >= 48) else 2 if (tiles <= 320 and nk >= 24) else 1)
            return (64 if (ns == 1 and tiles <= 160) else 128, 128, ns)
        floors = None
        try:
            floors = family_floors(args.precision, M_eff, H, I, B, L_eff, cfgd['num_attention_heads'],
                                   plan={'fp32x3': _plan_x3, 'bf16': _plan_b16}.get(args.precision), sq=sq)
        except Exception as e:                                   # noqa: BLE001 -- auxiliary: reported, never fatal
            prof_error = (prof_error or '') + ' floors: ' + repr(e)
        families = []
        for k, (name, bound, work) in fam.items():
            in_run = k_n[k] > 0
            n, tot, steps = (k_n[k], k_ms[k], args.steps) if in_run else (e_n[k], e_ms[k], EV_STEPS)
            if n == 0 or not tot > 0:          # nothing (or nothing sane) recorded for this family: leave it out
                continue
            sec = tot * 1e-3 / steps                   # seconds of this family per step
            pk = (peak_attn if name.startswith('attention') else peak) * 1e12 if bound == 'mfma' else 8.0e12
            ev_us = (e_ms[k] * 1e3 / e_n[k]) if (in_run and e_n[k] > 0 and e_ms[k] > 0) else None
            families.append({'family': name, 'bound': bound, 'launches_per_step': n // steps,
                             'avg_us': round(tot * 1e3 / n, 2), 'ms_per_step': round(sec * 1e3, 4),
                             'achieved': round(work / sec / 1e12, 2), 'unit': 'TFLOP/s' if bound == 'mfma' else 'TB/s',
                             'frac': round(work / sec / pk, 4), 'peak': round(pk / 1e12, 1) if bound == 'mfma' else 8.0, 'kernel': kernel_of[pk_name].get(k),
                             'measured': 'in-kernel stamps inside the timed region' if in_run else
                                         'HIP events, separate %d-step pass after the timed region (%.2f ms/step under events)' % (EV_STEPS, ev_ms)})
            if floors is not None and name in floors:
                fl = floors[name]
                fl_ms = fl['floor_us'] * nl * 1e-3
                families[-1]['floor'] = {'mfma_us': fl['mfma_us'], 'intake_us': fl['intake_us'], 'hbm_us': fl['hbm_us'], 'floor_us': fl['floor_us'],
                                         'per': 'layer (sum over the family\'s launches of one layer)', 'launches': fl['launches']}
                families[-1]['floor_ms_per_step'] = round(fl_ms, 4)
                families[-1]['over_floor'] = round(sec * 1e3 / fl_ms, 3) if fl_ms > 0 else None
            if ev_us is not None:
                # the same family under HIP events (the separate pass): from the moment the launch could start on its stream to
                # its end -- the time a dispatch sits queued behind the other stream's workgroups is IN this one, not in the stamps
                families[-1]['avg_us_dispatch'] = round(ev_us, 2)
                families[-1]['frac_dispatch'] = round(work / (ev_us * 1e-6 * (n // steps)) / pk, 4)
        out = {
            'metric': 'train samples/sec UNITER-%s (%d regions, %d tok)' % (args.model, R, T),
            'value': round(value, 2), 'unit': 'samples/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(ms, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': dt_name, 'data': 'synthetic',
            'config': {'workload': ('UNITER-%s fine-tune step (fwd + BCE + bwd + clip + Adam, dropout 0.1), '
                                    if args.workload == 'finetune' else
                                    'UNITER-%s + ITM/MLM/MRFR heads, one task per step (fwd + loss + bwd + clip + Adam, '
                                    'dropout 0.1; BASELINE configs[4]; FLOP fractions count the encoder only), ') % args.model +
                                   'batch %d per GPU, %d regions x 2048, %d text tokens%s, %s'
                                   % (B, R, T, (' (ragged: joint length %d%s)' % (L_eff, ', packed' if args.packed else '')) if args.ragged else '',
                                      'fp32 (BASELINE configs[1]; native fp32 MFMA kernels)' if args.precision == 'fp32' else
                                      'fp32 (BASELINE configs[1]); dense products by an exact 3 x bf16 operand split: 6 MFMA products '
                                      'per block, fp32 accumulate (error vs float64 <= the native fp32 MFMA kernel\'s: '
                                      'tests/test_gemm_x3_gpu.py); the attention\'s own products the same way (L <= 192, csrc/attention_x3.hip), softmax / LayerNorm / '
                                      'loss / optimizer fp32' if args.precision == 'fp32x3'
                                      else 'bf16 MFMA GEMMs / fp32 storage (BASELINE configs[2])'),
                       'global_batch': B * world, 'parallelism': 'dp%d' % world,
                       # what torch.distributed itself reports (not the flag): lets the driver verify the collective saw N ranks
                       'ranks_seen': dist.get_world_size() if use_dist else 1,
                       'dist_backend': dist.get_backend() if use_dist else None,
                       'launcher': ('bench.py (self-spawned ranks)' if os.environ.get('UNITER_BENCH_LAUNCHER') == 'self' else
                                    'external (RANK in the environment)') if use_dist else 'none (single process)',
                       'side_stream_wgrad': not args.no_side_stream,
                       'optimizer_overlaps_next_forward': not args.no_adam_overlap,
                       'grad_payload': sync.payload if sync is not None else None,
                       'dp_sparse_embedding_steps': sync.sparse_steps if sync is not None else 0,
                       'hip_hw_queues': os.environ.get('GPU_MAX_HW_QUEUES')},
            # the step's matrix work priced on the pipes it runs on (dense products at `peak`, attention at its own) over the step time
            'step_mfma_frac': round(((total - 3 * nl * att) / (peak * 1e12) + 3 * nl * att / (peak_attn * 1e12)) / (ms * 1e-3), 4),
            'ffn_roofline_frac': round(ffn / (ms * 1e-3) / (peak * 1e12), 4),
            'peak_tflops': {'dense_products': round(peak, 1), 'attention': round(peak_attn, 1),
                            'derivation': ('2500 (bf16 dense, MI355X_MICROARCH.md) / 6 products per fp32-equivalent block = 416.7'
                                           if args.precision == 'fp32x3' else 'MI355X_MICROARCH.md')},
            'final_loss': round(loss, 5),
        }
        if native is not None:
            out['native_fp32'] = native
        if bf16_leg is not None:
            out['bf16'] = bf16_leg
        if comm is not None:
            comm['headline_exchange'] = 'sparse word-embedding rows' if comm['sparse_embeddings'] else 'dense'
            out['comm'] = comm
        if prof_error:
            out['profiling_error'] = prof_error
        if families:
            # `roofline` = the GEMM family that takes the most time of the step (the weight- and input-gradient GEMMs
            # run on two streams beside each other: their in-situ durations include that sharing); every timed family
            # follows in `roofline_families`
            gemms = [f for f in families if f['family'].startswith('gemm_')]
            dom = max(gemms, key=lambda f: f['ms_per_step']) if gemms else families[0]
            build_info = lib.uniter_build_info().decode()
            out['roofline'] = {'bound': dom['bound'], 'achieved': dom['achieved'], 'peak': peak, 'unit': dom['unit'],
                               'frac': dom['frac'], 'traffic': None, 'kernel': '%s: %s' % (dom['family'], dom['kernel']),
                               'launches': dom['launches_per_step'] * args.steps, 'avg_ms': round(dom['avg_us'] * 1e-3, 4),
                               'avg_ms_dispatch': round(dom['avg_us_dispatch'] * 1e-3, 4) if 'avg_us_dispatch' in dom else None,
                               'frac_dispatch': dom.get('frac_dispatch'),
                               'note': '`frac` / `avg_ms` use the in-kernel stamps of the timed region (first workgroup start -> last workgroup end '
                                       'of each launch); `frac_dispatch` / `avg_ms_dispatch` are the same launches under HIP events in the separate '
                                       '%d-step pass (time queued behind the other stream included: what a rocprofv3 kernel trace reports)' % EV_STEPS,
                               'share_of_step_kernel_time': round(dom['ms_per_step'] / sum(f['ms_per_step'] for f in families), 3)}
            out['roofline_families'] = families
            if bwd_union_ms.value > 0:
                # the input- and weight-gradient GEMMs run beside each other on two streams: each family's in-situ rate
                # above includes that sharing; this is the rate of both together over the time either of them ran
                wk = 2 * fam[6][2]
                sec_u = bwd_union_ms.value * 1e-3 / args.steps
                out['backward_gemms_together'] = {'bound': 'mfma', 'ms_per_step': round(sec_u * 1e3, 4),
                                                  'achieved': round(wk / sec_u / 1e12, 2), 'unit': 'TFLOP/s', 'peak': peak,
                                                  'frac': round(wk / sec_u / (peak * 1e12), 4),
                                                  'measured': 'union of the stamped launch intervals of gemm_dgrad and gemm_wgrad inside the timed region'}
            tr = pmc_traffic(args, M_eff, cfgd, build_info)
            if tr is not None:
                out['traffic_from_profile'] = dict(tr, source='%s (rocprofv3 --pmc passes of this command on this library build; '
                                                              'FETCH_SIZE doubled per the gfx950 correction, WRITE_SIZE exact)' % tr['source'])
                if dom['family'] in tr:      # the dominant kernel's bytes per launch from those passes (same build only)
                    out['roofline']['traffic'] = tr[dom['family']]['bytes_per_launch']
                    out['roofline']['traffic_algorithmic'] = tr[dom['family']]['algorithmic_bytes']
                    out['roofline']['traffic_source'] = 'traffic_from_profile (counter passes of this command; not read inside the timed run)'
        # the optimizer step alone (HBM-bound: 32 B / parameter), measured after the timed region.  This and the CPU leg
        # below are auxiliary measurements: if one of them fails, the line of the timed region is still printed
        try:
            optimizer_alone(out, opt, model)
        except Exception as e:                                   # noqa: BLE001 -- reported, never silent
            out['optimizer_error'] = repr(e)
        if world == 1 and not args.no_cpu_baseline:
            try:
                out['cpu_baseline'] = cpu_baseline(all_cores=args.cpu_all_cores)
            except Exception as e:                               # noqa: BLE001
                out['cpu_baseline_error'] = repr(e)
        # step-level floor: sum of the families' floors (the two backward streams overlap: a serial sum is an upper bound of the true floor)
        if families and any('floor_ms_per_step' in f for f in families):
            out['step_floor_ms'] = round(sum(f.get('floor_ms_per_step', 0.0) for f in families), 4)
            out['step_over_floor'] = round(ms / out['step_floor_ms'], 3) if out['step_floor_ms'] > 0 else None
        out['attention_kernel'] = ('attn_x3_fwd_kernel / attn_x3_bwd_kernel (csrc/attention_x3.hip: the attention products as x3 products on the bf16 pipe)' if attn_x3 else
                                   'attn_b16x kernels (csrc/attention_x3.hip, one piece)' if args.precision == 'bf16' and L_eff <= 192 else
                                   'attn_fwd_split / attn_bwd_* (csrc/attention_f32.hip: fp32 MFMAs%s)' %
                                   ('; the x3 attention kernels hold L <= %d, this batch has L = %d' % (lib.uniter_attn_x3_max_len(), L_eff)
                                    if args.precision == 'fp32x3' else ''))
        # Order of the line: the long blocks first, the short figures a reader wants LAST -- a record that keeps
        # only the tail of the line still holds them; `summary` repeats them in a few hundred bytes at the very end
        order_last = ['roofline', 'cpu_baseline', 'optimizer', 'native_fp32', 'bf16', 'step_mfma_frac', 'ffn_roofline_frac',
                      'step_floor_ms', 'step_over_floor', 'attention_kernel', 'final_loss']
        for k_ in order_last:
            if k_ in out:
                out[k_] = out.pop(k_)
        summ = {'value': out['value'], 'ms_per_step': out['ms_per_step'], 'precision': args.precision,
                'step_mfma_frac': out.get('step_mfma_frac'), 'ffn_roofline_frac': out.get('ffn_roofline_frac'),
                'step_floor_ms': out.get('step_floor_ms'), 'step_over_floor': out.get('step_over_floor')}
        if isinstance(out.get('roofline'), dict):
            summ['roofline'] = {k_: out['roofline'].get(k_) for k_ in ('frac', 'achieved', 'peak', 'unit', 'traffic')}
            summ['roofline']['family'] = (out['roofline'].get('kernel') or '').split(':')[0]
        summ['over_floor'] = {f['family']: f.get('over_floor') for f in (out.get('roofline_families') or []) if 'over_floor' in f}
        for k_ in ('native_fp32', 'bf16'):
            if isinstance(out.get(k_), dict):
                summ[k_] = {a: out[k_].get(a) for a in ('value', 'ms_per_step', 'step_mfma_frac', 'error') if a in out[k_]}
        if isinstance(out.get('optimizer'), dict):
            summ['optimizer'] = {a: out['optimizer'].get(a) for a in ('ms', 'achieved', 'unit', 'frac')}
        if isinstance(out.get('cpu_baseline'), dict):
            summ['cpu_baseline'] = {a: out['cpu_baseline'].get(a) for a in ('value', 'unit', 'cores', 'kind')}
        out['summary'] = summ
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + '\n').encode())
    if use_dist:
        dist.destroy_process_group()
    return 0


def optimizer_alone(out, opt, model):
    """adam_kernel alone: 10 launches behind 2 warm-ups, every chunk on the update path (bytes / time against 8 TB/s)"""
    import torch
    opt.join()
    torch.cuda.synchronize()
    numel = model.param_store().numel
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    saved, opt.overlap_encoder = opt.overlap_encoder, None
    for it in range(12):
        if it == 2:
            e0.record()
        model.param_store().touch(model.param_store().names)       # every chunk takes the update path
        opt.step(grad_scale=1.0, max_grad_norm=0.0, zero_grads=True)
    e1.record()
    torch.cuda.synchronize()
    opt.overlap_encoder = saved
    o_ms = e0.elapsed_time(e1) / 10
    # read p, g, m, v + write p, m, v = 28 B per parameter.  The 10 timed launches find every gradient ZERO (the launch before
    # them cleared it) and adam_kernel stores the zero_grad clear only where something was written: no clear bytes here.  The
    # weight mirror of the bf16 / fp32x3 modes adds its 2 / 6 bytes per parameter (this measurement runs one launch over the whole
    # buffer: every parameter is mirrored; in the step only the encoder layers' are)
    st = model.param_store()
    mirror_b = 2 * getattr(st, 'mirror_pieces', 1) if getattr(st, 'mirror', None) is not None else 0
    nbytes = (28 + mirror_b) * numel
    out['optimizer'] = {'bound': 'hbm', 'bytes': nbytes, 'bytes_per_parameter': round(nbytes / numel, 2), 'ms': round(o_ms, 4),
                        'achieved': round(nbytes / (o_ms * 1e-3) / 1e12, 3), 'unit': 'TB/s', 'peak': 8.0,
                        'frac': round(nbytes / (o_ms * 1e-3) / 8.0e12, 4),
                        'note': 'adam_kernel alone, 10 launches (after 2 warm-ups) behind the timed region, every chunk on the update path, gradients already zero (no clear stores)%s' % (', + %d B / parameter of weight mirror' % mirror_b if mirror_b else '')}


_AS_SCRIPT = False

if __name__ == '__main__':
    _AS_SCRIPT = True
    sys.exit(main())
