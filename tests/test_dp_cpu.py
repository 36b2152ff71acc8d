"""CPU, world_size 2 over gloo: the bucketed gradient exchange (meme_challenge_amd/dp.py).
Invariant: summed-then-averaged per-rank gradients equal the
single-process gradients on the concatenated batch, every rank ends with identical
buffers, buckets are contiguous slices issued in backward order, and non-stepping
micro-batches do not communicate."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))

from common import TINY, TINY_IMG_DIM      # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _layout(sd):
    from meme_challenge_amd.model import ParamStore, CHUNK
    named = [(k, v) for k, v in sd.items()]
    order, buckets = ParamStore._order(named)
    offs, off = {}, 0
    for n, p in order:
        offs[n] = off
        off += (p.numel() + CHUNK - 1) // CHUNK * CHUNK
    ranges = []
    for names in buckets:
        s = min(offs[n] for n in names)
        e = max(offs[n] + (sd[n].numel() + CHUNK - 1) // CHUNK * CHUNK for n in names)
        ranges.append((s, e))
    return offs, off, ranges, buckets


def _grads(sd, batch):
    from oracle import uniter_oracle as O
    from oracle import step_oracle as S
    leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    kw = dict(img_feat=batch['img_feat'], img_pos_feat=batch['img_pos_feat'], input_ids=batch['input_ids'],
              position_ids=batch['position_ids'], attention_mask=batch['attn_mask'],
              gather_index=batch['gather_index'], output_all_encoded_layers=False)
    loss = S.bce_with_logits(O.meme_uniter_forward(leaf, TINY, **kw), batch['labels'], 1.8)
    loss.backward()
    return {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaf.items()}


def _worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from oracle import uniter_oracle as O
    from meme_challenge_amd.dp import GradSync
    torch.set_num_threads(2)
    sd = O.synth_state_dict(TINY, seed=3, img_dim=TINY_IMG_DIM, ln_jitter=0.05)
    offs, numel, ranges, buckets = _layout(sd)
    nl = TINY['num_hidden_layers']
    assert len(ranges) == nl + 2
    # buckets are contiguous, ordered head | layer nl-1 .. 0 | embeddings, and tile the buffer
    assert ranges[0][0] == 0 and ranges[-1][1] == numel
    for (s0, e0), (s1, e1) in zip(ranges, ranges[1:]):
        assert e0 == s1
    assert any('linear.weight' in n for n in buckets[0]) and any('layer.%d.' % (nl - 1) in n for n in buckets[1])
    assert any('word_embeddings' in n for n in buckets[-1])
    batch = O.synth_batch(3, 10, 6, seed=50 + rank, vocab=TINY['vocab_size'], img_dim=TINY_IMG_DIM,
                          txt_lens=[10, 5, 7], num_bbs=[6, 3, 6])
    g = _grads(sd, batch)
    flat = torch.zeros(numel)
    for n, t in g.items():
        flat[offs[n]:offs[n] + t.numel()] = t.reshape(-1)
    local = flat.clone()

    sync = GradSync(flat, ranges, bucket_bytes=1)       # every bucket its own collective
    # (1) a non-stepping micro-batch must not communicate
    sync.prepare(will_step=False)
    sync.hook('begin', None, None)
    for l in range(nl - 1, -1, -1):
        sync.hook('layer', l, None)
    sync.hook('embed', None, None)
    sync.finish()
    assert torch.equal(flat, local) and sync.launched == []
    # (2) the stepping micro-batch: bucketed all-reduce in backward order
    sync.prepare(will_step=True)
    sync.hook('begin', None, None)
    for l in range(nl - 1, -1, -1):
        sync.hook('layer', l, None)
    sync.hook('embed', None, None)
    sync.finish()
    assert sync.launched == ranges
    # (3) coalescing: head + all layers travel as ONE collective, flushed when layer 0 completes (not held back for the
    # embeddings), and the embeddings are a collective of their own
    flat2 = local.clone()
    sync2 = GradSync(flat2, ranges, bucket_bytes=1 << 40)
    sync2.prepare(True)
    sync2.hook('begin', None, None)
    for l in range(nl - 1, -1, -1):
        sync2.hook('layer', l, None)
        assert sync2.launched == ([] if l > 0 else [(0, ranges[-1][0])])
    sync2.hook('embed', None, None)
    assert sync2.launched == [(0, ranges[-1][0]), ranges[-1]]
    # per-block waiting (the optimizer without clipping): only the buckets that overlap the block are waited for
    sync2.wait_range(ranges[1][0], ranges[1][1])
    assert [r[3] for r in sync2._inflight] == [True, False]
    sync2.finish()
    assert [r[3] for r in sync2._inflight] == [True, True]
    assert torch.equal(flat2, flat)
    # (3b) bf16 payload: the slices are rounded to bf16, summed in bf16 and widened back; local buffer stays fp32
    flat4 = local.clone()
    sync4 = GradSync(flat4, ranges, bucket_bytes=1, payload='bf16')
    sync4.prepare(True)
    sync4.hook('begin', None, None)
    for l in range(nl - 1, -1, -1):
        sync4.hook('layer', l, None)
    sync4.hook('embed', None, None)
    sync4.finish()
    assert sync4.launched == ranges and flat4.dtype == torch.float32
    parts = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(parts, local)
    expect = sum(p_.to(torch.bfloat16) for p_ in parts[1:]) if world > 1 else 0
    expect = (parts[0].to(torch.bfloat16) + expect).to(torch.float32) if world > 1 else parts[0].to(torch.bfloat16).float()
    assert torch.equal(flat4, expect)
    assert (flat4 - flat).abs().max().item() <= 2.0 ** -7 * flat.abs().max().item()
    # (4) finish() without hooks (e.g. frozen encoder) still reduces everything
    flat3 = local.clone()
    sync3 = GradSync(flat3, ranges, bucket_bytes=1)
    sync3.prepare(True)
    sync3.finish()
    assert torch.equal(flat3, flat)
    # (5) sparse word-embedding exchange: the table leaves the dense collectives; (sorted ids, rows at first occurrences)
    # are all-gathered and summed in rank order over cleared rows -- bit-identical to the all-reduce for two ranks
    # (0 + a + b), identical on every rank for any world size, and the clip-norm pieces tile the buffer
    wname = 'uniter_model.embeddings.word_embeddings.weight'
    V, H = sd[wname].shape
    for payload in ('fp32', 'bf16'):
        flat5 = local.clone()
        sync5 = GradSync(flat5, ranges, bucket_bytes=1, payload=payload, word_table=(offs[wname], V, H))
        assert len(sync5.ranges) == len(ranges) + 1 and sync5.ranges[-2] == (offs[wname], offs[wname] + V * H)
        # micro-batch 1 accumulates locally (its ids are remembered), micro-batch 2 steps: the union of both is exchanged
        extra = torch.tensor([[3, 3, 96, 0]]) if rank == 0 else torch.tensor([[5, 1, 1, 0]])
        sync5.prepare(will_step=False, token_ids=extra)
        sync5.prepare(will_step=True, token_ids=batch['input_ids'])
        sync5.hook('begin', None, None)
        for l in range(nl - 1, -1, -1):
            sync5.hook('layer', l, None)
        sync5.hook('embed', None, None)
        assert sync5.sparse_steps == 1 and sync5.launched[-1] == sync5.ranges[-2]      # dense rest first, then the rows
        pieces = sorted(sync5.pieces())
        assert pieces[0][0] == 0 and pieces[-1][1] == numel and all(a[1] == b[0] for a, b in zip(pieces, pieces[1:]))
        sync5.finish()
        if payload == 'fp32':
            # rows no rank announced are untouched by the exchange: they hold the LOCAL gradient -- zero, as nobody looked them up
            assert torch.equal(flat5, flat), (flat5 - flat).abs().max()
        else:
            # bf16 rows summed in fp32: the table holds the unrounded sum, its bf16 copy (what the fused optimizer reads) is
            # what the dense bf16 all-reduce leaves (two ranks: one rounding of an exact sum either way)
            w0, w1 = offs[wname], offs[wname] + V * H
            assert torch.equal(flat5[:w0], flat4[:w0]) and torch.equal(flat5[w1:], flat4[w1:])
            assert torch.equal(flat5[w0:w1].to(torch.bfloat16).float(), flat4[w0:w1])
            assert torch.equal(sync5.comm[w0:w1].float(), flat4[w0:w1])
        chk5 = [torch.zeros_like(flat5) for _ in range(world)]
        dist.all_gather(chk5, flat5)
        assert all(torch.equal(c, chk5[0]) for c in chk5)
        # a step whose table gradient is dense (the MLM task's tied decoder) announces no ids: dense path, same sums
        flat6 = local.clone()
        sync6 = GradSync(flat6, ranges, bucket_bytes=1, payload=payload, word_table=(offs[wname], V, H))
        sync6.prepare(will_step=True, token_ids=None)
        sync6.hook('begin', None, None)
        for l in range(nl - 1, -1, -1):
            sync6.hook('layer', l, None)
        sync6.hook('embed', None, None)
        sync6.finish()
        assert sync6.sparse_steps == 0 and torch.equal(flat6, flat if payload == 'fp32' else flat4)
        assert sync6.launched == sync6.ranges
    # (7) sparse exchange under gradient accumulation: the reference's trainer steps after ONE micro-batch at
    # iteration 0 and after `accum` of them from then on, so the second exchange carries twice the ids of the first -- the
    # capacity agreed at the first one must hold a full window (dp.attach(accum=...))
    flat7 = local.clone()
    sync7 = GradSync(flat7, ranges, bucket_bytes=1, word_table=(offs[wname], V, H), accum=2)

    def one_exchange(sync, micro):
        for k, ids in enumerate(micro):
            sync.prepare(will_step=k == len(micro) - 1, token_ids=ids)
        sync.hook('begin', None, None)
        for l in range(nl - 1, -1, -1):
            sync.hook('layer', l, None)
        sync.hook('embed', None, None)
        sync.finish()
    one_exchange(sync7, [batch['input_ids']])                              # iteration 0: one micro-batch
    one_exchange(sync7, [batch['input_ids'], batch['input_ids'] + 1])      # a full window of two
    assert sync7.sparse_steps == 2 and sync7._cap == 2 * batch['input_ids'].numel()
    sync8 = GradSync(local.clone(), ranges, bucket_bytes=1, word_table=(offs[wname], V, H))      # accum not announced
    one_exchange(sync8, [batch['input_ids']])
    try:
        one_exchange(sync8, [batch['input_ids'], batch['input_ids'] + 1])
        raise AssertionError('expected the capacity error')
    except ValueError as e:
        assert 'gradient_accumulation' in str(e)
    # a sync that cannot exchange (one rank, no UNITER_DP_FORCE) remembers no token ids (they would pile up for the whole run)
    if rank == 0:
        lone = GradSync(local.clone(), ranges, bucket_bytes=1, word_table=(offs[wname], V, H))
        lone.world = 1
        for _ in range(3):
            lone.prepare(will_step=True, token_ids=batch['input_ids'])
        assert lone._tokens == [] and not lone.active
    if rank == 0:
        torch.save({'reduced': flat, 'offs': offs}, out)
    # every rank holds identical reduced gradients
    chk = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(chk, flat)
    assert all(torch.equal(c, chk[0]) for c in chk)
    dist.destroy_process_group()


def test_bucketed_allreduce_equals_big_batch_gradients(tmp_path):
    from oracle import uniter_oracle as O
    world, port, out = 2, _free_port(), str(tmp_path / 'r0.pt')
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    r = torch.load(out)
    sd = O.synth_state_dict(TINY, seed=3, img_dim=TINY_IMG_DIM, ln_jitter=0.05)
    bs = [O.synth_batch(3, 10, 6, seed=50 + k, vocab=TINY['vocab_size'], img_dim=TINY_IMG_DIM,
                        txt_lens=[10, 5, 7], num_bbs=[6, 3, 6]) for k in range(world)]
    big = {k: torch.cat([b[k] for b in bs], 0) for k in bs[0]}
    g = _grads(sd, big)
    for n, t in g.items():
        o = r['offs'][n]
        got = r['reduced'][o:o + t.numel()].view(t.shape) / world      # grad_scale = 1/world
        assert (got - t).abs().max().item() <= 1e-6 + 1e-4 * t.abs().max().item(), n


def test_cu_reserve_default_and_rccl_env(monkeypatch):
    """CUs the persistent matrix kernels leave to the exchange's kernels: 16 with more than one rank, none on one rank,
    UNITER_DP_CU_RESERVE overrides; RCCL's channels are capped at the reserve only on request (UNITER_DP_CAP_CHANNELS=1) and never
    against the caller's own NCCL_MAX_NCHANNELS."""
    from meme_challenge_amd import dp
    monkeypatch.delenv('UNITER_DP_CU_RESERVE', raising=False)
    assert dp.cu_reserve_default(1) == 0 and dp.cu_reserve_default(8) == dp.DEFAULT_CU_RESERVE == 16
    monkeypatch.setenv('UNITER_DP_CU_RESERVE', '24')
    assert dp.cu_reserve_default(1) == 24 and dp.cu_reserve_default(8) == 24
    monkeypatch.setenv('UNITER_DP_CU_RESERVE', '0')
    assert dp.cu_reserve_default(8) == 0
    monkeypatch.setenv('UNITER_DP_CU_RESERVE', 'junk')
    assert dp.cu_reserve_default(8) == 0
    monkeypatch.delenv('UNITER_DP_CU_RESERVE', raising=False)
    env = {}
    assert dp.prepare_rccl_env(8, env) == 16 and env == {}                                # RCCL's channel count is RCCL's by default
    env = {'UNITER_DP_CAP_CHANNELS': '1'}
    assert dp.prepare_rccl_env(8, env) == 16 and env['NCCL_MAX_NCHANNELS'] == '16'       # on request: no more channels than reserved CUs
    env = {'UNITER_DP_CAP_CHANNELS': '1', 'NCCL_MAX_NCHANNELS': '32'}
    assert dp.prepare_rccl_env(8, env) == 16 and env['NCCL_MAX_NCHANNELS'] == '32'       # the caller's choice stands
    env = {}
    assert dp.prepare_rccl_env(1, env) == 0 and env == {}


def _pick_worker(rank, world, port, out):
    import time
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from meme_challenge_amd import dp

    class Enc(object):
        cu_reserve = 16
    flat = torch.zeros(256)
    gs = dp.GradSync(flat, [(0, 128), (128, 256)])
    gs.cu_reserve = 16
    enc = Enc()
    seen = []

    def step():                       # reserve 0 is the faster setting on rank 0 but far slower on rank 1: the slowest rank decides
        seen.append(enc.cu_reserve)
        time.sleep({16: 0.004, 0: 0.001 if rank == 0 else 0.012}[enc.cu_reserve])
    r1 = dp.pick_cu_reserve(gs, enc, step, candidates=[16, 0], steps=3, warm=1)
    after1 = (enc.cu_reserve, gs.cu_reserve, list(seen))

    def step2():                      # the default candidates: the reserve in force, none, the default, three times the default
        time.sleep({16: 0.008, 0: 0.001, 48: 0.004}[enc.cu_reserve])
    r2 = dp.pick_cu_reserve(gs, enc, step2, steps=3, warm=1)
    after2 = (enc.cu_reserve, gs.cu_reserve)

    def step3():                      # a failure every rank meets alike: the reserve in force stays, the result says why
        if enc.cu_reserve == 16:
            raise RuntimeError('workspace too small')
    r3 = dp.pick_cu_reserve(gs, enc, step3, candidates=[0, 16], steps=2, warm=1)
    after3 = (enc.cu_reserve, gs.cu_reserve)
    # round 6: a wall-time budget -- the slowest rank's clock decides, every rank skips the same candidates
    enc.cu_reserve = gs.cu_reserve = 16

    def step4():
        time.sleep(0.02 if rank == 1 else 0.001)
    r4 = dp.pick_cu_reserve(gs, enc, step4, candidates=[16, 0, 48], steps=2, warm=1, budget_s=0.1)
    # round 6: the sparse exchange sized statically (batch_size x max_txt_len x accumulation): no agreement collective, and a later
    # batch with more ids than the first fits
    flat2 = torch.zeros(64 + 16 * 8 + 64)
    gss = dp.GradSync(flat2, [(0, 64), (64, 64 + 16 * 8 + 64)], word_table=(64, 16, 8), accum=2, token_capacity=6)
    cap0 = (gss._cap, gss.cap_source)
    gss.prepare(True, token_ids=torch.tensor([1, 2, 3]))               # the first exchange: 3 ids
    n1 = gss._sorted[0].numel()
    gss.flush_all()
    gss.prepare(False, token_ids=torch.tensor([4, 5, 6, 7, 8, 9]))     # a later window: 6 + 5 = 11 ids <= 12
    gss.prepare(True, token_ids=torch.tensor([1, 1, 2, 3, 5]))
    n2 = gss._sorted[0].numel()
    gss.flush_all()
    try:
        gss.prepare(True, token_ids=torch.arange(13))
        too_many = None
    except ValueError as e:
        too_many = str(e)
    torch.save(dict(r1=r1, after1=after1, r2=r2, after2=after2, r3=r3, after3=after3, r4=r4, cap0=cap0, n1=n1, n2=n2, too_many=too_many),
               out + str(rank))
    dist.destroy_process_group()


def test_cu_reserve_is_picked_by_the_slowest_rank(tmp_path):
    """dp.pick_cu_reserve: every rank times the same candidates, the maximum over ranks decides, all ranks end with the same
    reserve; a failure leaves the reserve as it was.  One rank (or no exchange): no-op."""
    from meme_challenge_amd import dp
    assert dp.pick_cu_reserve(None, None, lambda: None) is None
    world, port, out = 2, _free_port(), str(tmp_path / 'pick')
    mp.spawn(_pick_worker, args=(world, port, out), nprocs=world, join=True)
    rs = [torch.load(out + str(r)) for r in range(world)]
    for r in rs:
        assert r['r1']['picked'] == 16 and [c['cu_reserve'] for c in r['r1']['candidates']] == [16, 0]
        assert r['after1'][:2] == (16, 16) and r['after1'][2] == [16] * 4 + [0] * 4
        assert r['r1']['candidates'][1]['ms_per_step'] > r['r1']['candidates'][0]['ms_per_step']
        assert r['r2']['picked'] == 0 and r['after2'] == (0, 0) and [c['cu_reserve'] for c in r['r2']['candidates']] == [16, 0, 48]
        # (the third call starts from reserve 0: candidates 0, then 16, which raises)
        assert r['r3']['picked'] == 0 and 'workspace too small' in r['r3']['error'] and r['after3'] == (0, 0)
        # budget 0.1 s: the first candidate takes rank 1 0.06 s (3 steps of 20 ms) -- the second would not fit: skipped everywhere
        assert [c['cu_reserve'] for c in r['r4']['candidates']] == [16] and r['r4']['skipped'] == [0, 48] and r['r4']['picked'] == 16
        assert r['r4']['wall_s'] >= 0.05 and r['r4']['budget_s'] == 0.1 and 'wall_s' in r['r1'] and 'skipped' not in r['r1']
        assert r['cap0'] == (12, 'static') and r['n1'] == 12 and r['n2'] == 12
        assert r['too_many'] and 'token_capacity' in r['too_many']
    assert rs[0]['r4']['wall_s'] == rs[1]['r4']['wall_s']
    rs0 = {k: v for k, v in rs[0]['r1'].items() if k != 'wall_s'}
    assert rs0 == {k: v for k, v in rs[1]['r1'].items() if k != 'wall_s'}
    assert {k: v for k, v in rs[0]['r2'].items() if k != 'wall_s'} == {k: v for k, v in rs[1]['r2'].items() if k != 'wall_s'}
    # the design's arithmetic for the exposed collective: 98 MB fp32 at 8 ranks on one link / on seven
    assert abs(dp.predicted_exposed_ms(97.6e6, 8, 1) - 2 * 7 / 8 * 97.6e6 / 153e9 * 1e3) < 1e-9
    assert abs(dp.predicted_exposed_ms(97.6e6, 8, 7) * 7 - dp.predicted_exposed_ms(97.6e6, 8, 1)) < 1e-9
    assert dp.predicted_exposed_ms(97.6e6, 1) == 0.0
