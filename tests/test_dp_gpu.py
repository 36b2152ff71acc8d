"""GPU: the data-parallel path on real hardware -- the model's backward hooks, dp.GradSync and an RCCL
all-reduce (backend 'nccl', one rank, UNITER_DP_FORCE=1) in a child process (a process group cannot be
created inside the pytest process without leaking into the other tests).

With one rank the sum is the identity, so two training steps through the collective path must leave
the parameters of the plain steps (fp32 payload) up to the summation order of the float-atomic weight
gradients -- the same 1e-9 a plain re-run shows --, the buckets must be the backward-order slices
(head + layers coalesced, flushed at layer 0; the embeddings alone), each collective must have been
issued on the stream that produced its gradients, and the bf16 payload must change the result by no more than
the rounding of the gradients (A16 / E1 of SURVEY.md section 8; reference mechanism train_template.py:58-59)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_single_rank_rccl_path_matches_plain_step():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', MASTER_ADDR='127.0.0.1', MASTER_PORT='29533')
    env.pop('RANK', None)
    r = subprocess.run([sys.executable, os.path.join(REPO, 'tests', 'tools', 'dp_single_rank_check.py')],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('DPCHECK ')][-1]
    out = json.loads(line[len('DPCHECK '):])
    noise = max(out['plain_rerun_maxdiff'], 2e-9)          # two plain runs differ by this much (fp32 atomics order)
    assert out['fp32_clip_maxdiff'] <= 4 * noise, out
    assert out['fp32_noclip_maxdiff'] <= 4 * noise, out     # per-block waits (no clipping) instead of one finish()
    assert out['loss_diff'] <= 1e-6, out
    info = out['info']
    ranges = [tuple(r) for r in info['ranges']]
    launched = [tuple(r) for r in info['launched']]
    # head | layer nl-1 .. 0 coalesced into one collective (the tiny model is far below the bucket size) flushed when
    # layer 0 completes, then the embeddings alone
    assert launched == [(ranges[0][0], ranges[-1][0]), ranges[-1]], (launched, ranges)
    # the layers' collective behind the weight-gradient kernels (side stream), the embeddings' behind the embedding
    # backward (main stream)
    assert info['on_side_stream'] == [True, False], info['on_side_stream']
    assert out['bf16_info']['payload'] == 'bf16'
    # bf16 payload: gradients rounded once (2^-9 relative) before Adam; after two steps at lr 1e-3 the parameters move by
    # at most a few lr relative to the run without the collective
    assert out['bf16_payload_maxdiff'] < 5e-3, out
    # the sparse word-embedding exchange through RCCL's all-gather (one rank: the gathered rows are the rank's own): the table
    # leaves the dense collectives (one more range: table | rest of the embeddings), same parameters as the plain steps
    sp = out['sparse_info']
    assert sp['sparse_steps'] == 2 and len(sp['sync_ranges']) == len(sp['ranges']) + 1, sp
    assert out['fp32_sparse_maxdiff'] <= 4 * noise, out
    assert out['bf16_sparse_steps'] == 2 and out['bf16_sparse_maxdiff'] < 5e-3, out


def test_two_ranks_share_the_batch():
    """Two processes (gloo, both on cuda:0) run the HIP backward with its hooks + GradSync on half a ragged batch each:
    the exchanged, averaged gradients equal the gradients of the whole batch in one process (fp32 to summation order; in
    the bf16 mode to the mode's own rounding), the mean of the shard losses equals the batch loss."""
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29541', WORLD_SIZE='2', HSA_ENABLE_IPC_MODE_LEGACY='0')
    script = os.path.join(REPO, 'tests', 'tools', 'dp_two_rank_check.py')
    procs = [subprocess.Popen([sys.executable, script], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, so[-2000:] + se[-4000:]
    line = [l for l in outs[0][0].splitlines() if l.startswith('DP2CHECK ')][-1]
    out = json.loads(line[len('DP2CHECK '):])
    f32, b16 = out['fp32'], out['bf16']
    assert f32['buckets'] >= 2
    # measured: 3e-8 absolute / 8e-8 rms-relative in both modes.  A sample's forward and input gradients do not depend on
    # which other samples share its batch (so the bf16 mode rounds exactly the same values on both sides); only the sums
    # over the batch -- weight and bias gradients, here split in two and added by the collective -- change their order
    # bf16 payload: every rank's gradients rounded to bf16 (2^-9 relative) before the sum
    assert out['bf16_payload']['rel_rms'] < 4e-3 and out['bf16_payload']['rel_rms'] > 1e-5, out['bf16_payload']
    # sparse word-embedding exchange (ids + touched rows all-gathered, summed in rank order): the same gradients
    sp, sp16 = out['fp32_sparse'], out['bf16_payload_sparse']
    assert sp['sparse_steps'] == 1 and sp16['sparse_steps'] == 1 and f32['sparse_steps'] == 0
    assert sp16['rel_rms'] < 4e-3, sp16
    for r in (f32, b16, sp):
        assert r['maxdiff'] <= 1e-6 * max(r['scale'], 1.0), r
        assert r['rel_rms'] < 1e-6, r
        assert abs(r['loss_mean'] - r['loss_ref']) < 1e-6, r
