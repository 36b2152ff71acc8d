"""uniter_gather_rows_ex: the joint rows of model/model.py:327-334 and their operand copy for the first encoder product from ONE launch,
against the two launches it replaces (uniter_gather_rows, then uniter_split3 / uniter_cast_bf16): bit-equal."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _L():
    from meme_challenge_amd import _lib
    return _lib


@pytest.mark.parametrize('B,S,Lout,H', [(16, 164, 164, 768), (3, 20, 17, 768), (2, 9, 9, 64), (5, 40, 33, 1024), (1, 4, 3, 264)])
@pytest.mark.parametrize('mode', [1, 2])
@pytest.mark.parametrize('with_index', [True, False])
def test_gather_with_operand_copy_equals_the_two_launches(B, S, Lout, H, mode, with_index):
    L = _L()
    lib, ptr, cs = L.lib(), L.ptr, L.cur_stream()
    if mode == 1 and H % 8:
        pytest.skip('uniter_split3 takes 8 columns per thread')
    g = torch.Generator().manual_seed(B * 7 + H + mode)
    cat = (torch.randn(B, S, H, generator=g) * torch.logspace(-3, 3, H)).cuda()
    gi = torch.randint(0, S, (B, Lout), generator=g).cuda() if with_index else None
    out0 = torch.empty(B, Lout, H, device='cuda')
    L.check(lib.uniter_gather_rows(ptr(cat), ptr(gi), ptr(out0), B, S, Lout, H, cs))
    if mode == 1:
        ref = torch.zeros(B * Lout, 3, H, dtype=torch.int16, device='cuda')
        L.check(lib.uniter_split3(ptr(out0), B * Lout, H, H, ptr(ref), 3 * H, H, cs))
        got = torch.full((B * Lout, 3, H), -1, dtype=torch.int16, device='cuda')
    else:
        ref = torch.zeros(B * Lout, H, dtype=torch.int16, device='cuda')
        L.check(lib.uniter_cast_bf16(ptr(out0), ptr(ref), out0.numel(), cs))
        got = torch.full((B * Lout, H), -1, dtype=torch.int16, device='cuda')
    out1 = torch.full((B, Lout, H), float('nan'), device='cuda')
    L.check(lib.uniter_gather_rows_ex(ptr(cat), ptr(gi), ptr(out1), ptr(got), mode, B, S, Lout, H, cs))
    torch.cuda.synchronize()
    assert torch.equal(out1, out0)
    if with_index:
        assert torch.equal(out0, torch.gather(cat, 1, gi[:, :, None].expand(B, Lout, H)))
    assert torch.equal(got, ref)


def test_gather_with_operand_copy_refuses_what_it_does_not_cover():
    L = _L()
    lib = L.lib()
    x = torch.zeros(4, 2048, device='cuda')
    o = torch.zeros(4, 3, 2048, dtype=torch.int16, device='cuda')
    assert lib.uniter_gather_rows_ex(L.ptr(x), None, L.ptr(x), L.ptr(o), 1, 1, 4, 4, 2048, None) != 0      # H > 1024
    assert lib.uniter_gather_rows_ex(L.ptr(x), None, L.ptr(x), L.ptr(o), 3, 1, 4, 4, 512, None) != 0       # unknown mode
    assert lib.uniter_gather_rows_ex(L.ptr(x), None, L.ptr(x), None, 1, 1, 4, 4, 512, None) != 0
