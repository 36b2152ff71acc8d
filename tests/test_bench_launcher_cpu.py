"""CPU: `python bench.py --gpus N` outside a launcher starts N fresh rank processes itself (bench.launch_ranks): each
child gets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT, only rank 0's stdout reaches the caller's stdout
(ONE JSON line), a failing rank stops the others and makes the launcher exit non-zero.  The parent makes no GPU call:
it is exercised here with stand-in children (the real ranks need an MI355X: tests/test_bench_gpu.py)."""
import json
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

CHILD = r'''
import json, os, sys, time
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
assert os.environ['LOCAL_RANK'] == os.environ['RANK'] and os.environ['MASTER_ADDR'] == '127.0.0.1'
assert int(os.environ['MASTER_PORT']) > 0 and os.environ['UNITER_BENCH_LAUNCHER'] == 'self'
assert 1 <= int(os.environ['OMP_NUM_THREADS']) <= 8 and os.environ['MKL_NUM_THREADS'] == os.environ['OMP_NUM_THREADS']      # (round 6: host pools capped per rank)
mode = sys.argv[1]
if mode == 'fail' and rank == 1:
    sys.exit(7)
if mode == 'fail':
    time.sleep(60)          # would hang in a collective: the launcher must stop it
print(json.dumps({'rank': rank, 'n_gpus': world, 'port': os.environ['MASTER_PORT']}), flush=True)
'''


def _run(mode, n):
    code = ('import sys; sys.path.insert(0, %r); import bench; '
            'sys.exit(bench.launch_ranks(%d, [sys.executable, "-c", %r, %r]))' % (REPO, n, CHILD, mode))
    return subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=120)


def test_launcher_starts_n_ranks_and_relays_rank0():
    out = _run('ok', 3)
    assert out.returncode == 0, out.stderr
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith('{')]
    assert lines == [{'rank': 0, 'n_gpus': 3, 'port': lines[0]['port']}]          # rank 0 only on stdout
    others = sorted(json.loads(l)['rank'] for l in out.stderr.splitlines() if l.startswith('{'))
    assert others == [1, 2]                                                         # the rest went to stderr


def test_launcher_fails_when_a_rank_fails_and_stops_the_others():
    t0 = time.time()
    out = _run('fail', 2)
    assert out.returncode == 7 and 'rank 1 exited with code 7' in out.stderr
    assert time.time() - t0 < 45          # rank 0 (sleeping for 60 s) was terminated, not waited for


def test_single_gpu_run_stays_in_process(monkeypatch):
    """N = 1 (the driver's BENCH line) never goes through the launcher; neither does a rank started by torch.distributed.run."""
    import bench
    called = []
    monkeypatch.setattr(bench, 'launch_ranks', lambda *a, **k: called.append(a) or 0)
    monkeypatch.setattr(bench, 'run_rank', lambda args: 0)
    monkeypatch.delenv('RANK', raising=False)
    assert bench.main(['--gpus', '1']) == 0 and called == []
    assert bench.main(['--gpus', '4', '--steps', '3']) == 0
    assert called and called[0][0] == 4 and called[0][1][-4:] == ['--gpus', '4', '--steps', '3']
    called.clear()
    monkeypatch.setenv('RANK', '2')
    assert bench.main(['--gpus', '4']) == 0 and called == []


def test_comm_block_schema_and_rccl_log_parse(tmp_path):
    """The `comm` block bench.py --gpus N adds to its line (per collective: bytes, issue -> done, exposed; payload; RCCL's own
    report) from the records dp.GradSync keeps, and the parser of rank 0's NCCL_DEBUG=INFO log."""
    import bench

    class FakeSync:
        payload, world, sparse_steps, last_sparse_rows = 'bf16', 8, 0, None

        def __init__(self):
            self.timings = [dict(start=0, end=1000, bytes=2000, issue_to_done_ms=1.5, exposed_ms=0.0),
                            dict(start=0, end=1000, bytes=2000, issue_to_done_ms=2.5, exposed_ms=0.5),
                            dict(start=1000, end=5000, bytes=8000, issue_to_done_ms=0.75, exposed_ms=0.75),
                            dict(start=1000, end=5000, bytes=8000, issue_to_done_ms=1.25, exposed_ms=0.25)]

        def collect_timings(self):
            return []
    log = tmp_path / 'rccl.log'
    log.write_text('host:1:1 [0] NCCL INFO Channel 00/08 :    0   1   2   3\nhost:1:1 [0] NCCL INFO Channel 07/08 :    0   1\n'
                   'host:1:1 [0] NCCL INFO Trees [0] -1/-1/-1->0->1\nhost:1:1 [0] NCCL INFO Connected all rings\n'
                   'host:1:1 [0] NCCL INFO AllReduce: 4194304 Bytes -> Algo RING proto SIMPLE channel{Lo..Hi}={0..7}\n')
    c = bench.comm_block(FakeSync(), 2, str(log))
    assert c['payload'] == 'bf16' and c['world'] == 8 and len(c['collectives']) == 2
    a, b = c['collectives']
    assert a['bytes'] == 2000 and a['per_step'] == 1.0 and a['issue_to_done_ms'] == 2.0 and a['exposed_ms'] == 0.25
    assert b['elements'] == 4000 and b['issue_to_done_ms'] == 1.0 and b['exposed_ms'] == 0.5
    assert c['bytes_per_step'] == 10000 and c['exposed_ms_per_step'] == 0.75
    # round 6: the design's arithmetic for the collective issued last, beside the measurement
    pe = c['predicted_exposed_ms']
    assert pe['last_collective_bytes'] == 8000 and pe['one_link_ring'] == round(2 * 7 / 8 * 8000 / 153e9 * 1e3, 4) and pe['all_links'] <= pe['one_link_ring']
    r = c['rccl']
    assert r['channels'] == 8 and r['algorithm'] == 'RING' and r['protocol'] == 'SIMPLE' and r['log_lines'] == 5
    assert bench.parse_rccl_log(str(tmp_path / 'missing.log'))['channels'] is None
    # physical cores of the CPU baseline: at least one, never more than the logical count
    n = bench._physical_cores()
    assert 1 <= n <= (os.cpu_count() or 1)


def test_round5_flags_parse_and_defaults():
    """The second timed regions are opt-in: by default `value` is ONE region, the requested exchange with the default CU reserve."""
    import bench
    a = bench.parse_args([])
    assert not a.both_exchanges and not a.reserve_ab and not a.no_bf16_leg and not a.cpu_all_cores and a.precision == 'fp32x3'
    assert not a.no_reserve_pick            # N > 1 measures the CU reserve before the warm-up unless told not to
    a = bench.parse_args(['--both_exchanges', '--reserve_ab', '--no_bf16_leg', '--cpu_all_cores', '--gpus', '8'])
    assert a.both_exchanges and a.reserve_ab and a.no_bf16_leg and a.cpu_all_cores and a.gpus == 8


def test_rank_host_threads_and_floor_arithmetic():
    """Round 6: (i) host threads per rank of a self-launched N-rank run; (ii) the floor bench.py prints per family,
    recomputed by hand for one launch: FFN-up forward of configs[1] on 128 x 256 x3 tiles."""
    import bench
    assert bench.rank_host_threads(1) >= 1 and bench.rank_host_threads(10 ** 6) == 1 and bench.rank_host_threads(8) <= 8
    a = bench.parse_args([])
    assert a.reserve_pick_budget_s == 20.0
    M, N, K = 2624, 3072, 768
    f = bench.gemm_floor(M, N, K, 128, 256, 32, 3, 1, 256, hbm_bytes=M * K * 6 + N * K * 6 + M * N * 10, flops=2.0 * M * N * K, peak_tflops=2500.0 / 6.0)
    tiles = 21 * 12
    assert f['tiles'] == tiles and f['items'] == tiles and f['cus_used'] == 252
    staged = tiles * 24 * (128 + 256) * 32 * 2 * 3
    assert abs(f['intake_us'] - staged / (70e9 * 252) * 1e6) < 0.01 and abs(f['mfma_us'] - 2.0 * M * N * K / (2500e12 / 6) * 1e6) < 0.01
    assert abs(f['hbm_us'] - (M * K * 6 + N * K * 6 + M * N * 10) / 8e12 * 1e6) < 0.01
    assert f['floor_us'] == max(f['mfma_us'], f['intake_us'], f['hbm_us'])
    # k-pieces: the same staged bytes over twice the work items (more CUs used)
    g1 = bench.gemm_floor(M, 768, 3072, 128, 128, 32, 3, 1, 256, 0, 1.0, 1.0)
    g2 = bench.gemm_floor(M, 768, 3072, 128, 128, 32, 3, 2, 256, 0, 1.0, 1.0)
    assert g1['cus_used'] == 126 and g2['cus_used'] == 252 and abs(g1['intake_us'] - 2 * g2['intake_us']) < 0.02
    fam = bench.family_floors('fp32x3', M, 768, 3072, 16, 164, 12)
    assert set(fam) >= {'gemm_ffn_up_fwd', 'gemm_dgrad', 'gemm_wgrad', 'attention_bwd', 'layernorm_fwd'}
    assert len(fam['gemm_dgrad']['launches']) == 4 and fam['gemm_dgrad']['floor_us'] >= fam['gemm_dgrad']['mfma_us']
    assert bench.family_floors('fp32', M, 768, 3072, 16, 164, 12)['gemm_ffn_up_fwd']['intake_us'] == 0.0      # no LDS-DMA staging there
