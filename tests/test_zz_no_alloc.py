"""GPU: sgpo_guidance_ratio_mlp_absorbing_noalloc (the entry point for callers that cannot allocate) against
sgpo_guidance_ratio_mlp_absorbing, through the C ABI: bit-exact log-ratios on two calls in a row, the second
with a smaller input; the size limit is refused with a clear error.  Collected last on purpose (the round's
end runs pytest -x)."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

MASK, K = 28, 31


def _case(gen, B, Lq):
    x = torch.randint(0, 26, (B, Lq), generator=gen)
    frac = torch.rand(B, 1, generator=gen)          # rows from fully masked to fully unmasked
    frac[0] = 1.0
    if B > 1:
        frac[1] = 0.0
    x[torch.rand(B, Lq, generator=gen) < frac] = MASK
    return x.cuda()


def _with_workspace(lib, L, vf, x, t):
    B, Lq = x.shape
    need = lib.sgpo_mlp_workspace_bytes(vf.handle, B)
    ws = torch.empty(need + 256, dtype=torch.uint8, device=x.device)
    base = (ws.data_ptr() + 255) & ~255
    out = torch.full((B, vf.n_read, K), float('nan'), dtype=torch.float32, device=x.device)
    L.check(lib.sgpo_guidance_ratio_mlp_absorbing(vf.handle, L.ptr(x), t, B, Lq, K, MASK, L.ptr(out), C.c_void_p(base),
                                                  need, L.stream_ptr()), 'sgpo_guidance_ratio_mlp_absorbing')
    return out


def _no_alloc(lib, L, vf, x, t, out):
    B, Lq = x.shape
    L.check(lib.sgpo_guidance_ratio_mlp_absorbing_noalloc(vf.handle, L.ptr(x), t, B, Lq, K, MASK, L.ptr(out),
                                                          L.stream_ptr()), 'sgpo_guidance_ratio_mlp_absorbing_noalloc')
    return out


def test_no_alloc_entry_point_matches_workspace_entry_point():
    from sgpo_amd import _lib as L
    from sgpo_amd import runtime as rt
    from sgpo_amd import weights as W
    rt.init('cuda:0')
    lib = L.load()
    assert lib.sgpo_guidance_noalloc_max_bytes() == 4 << 20
    gen = torch.Generator().manual_seed(11)
    for Lq, read_pos in ((119, list(range(119))), (389, [104, 105, 106, 117, 118, 162, 166, 182, 183, 184, 185, 227, 228, 230, 231])):
        vf = rt.MlpValueFunction(W.mlp_synthetic_state(seed=3, n_residues=len(read_pos)), read_pos)
        # big call first (fills most of the array), then a smaller one that finds the big call's leftovers
        # there, then the big one again; outputs are preallocated so that the calls themselves allocate nothing
        sizes = (2048, 37, 2048) if Lq == 119 else (1500, 3, 700)
        xs = [_case(gen, B, Lq) for B in sizes]
        outs = [torch.full((B, vf.n_read, K), float('nan'), dtype=torch.float32, device='cuda') for B in sizes]
        torch.cuda.synchronize()
        n_alloc = torch.cuda.memory_stats()['allocation.all.allocated']     # torch's count of allocations so far
        free_before = torch.cuda.mem_get_info()[0]                          # the driver's view: a hipMalloc would lower it
        for x, out in zip(xs, outs):
            _no_alloc(lib, L, vf, x, 5.0, out)
        torch.cuda.synchronize()
        assert torch.cuda.memory_stats()['allocation.all.allocated'] == n_alloc
        assert torch.cuda.mem_get_info()[0] == free_before
        for x, out in zip(xs, outs):
            ref = _with_workspace(lib, L, vf, x, 5.0)
            assert not bool(torch.isnan(ref).any())
            assert torch.equal(out, ref)


def test_no_alloc_entry_point_refuses_sizes_beyond_its_buffer():
    from sgpo_amd import _lib as L
    from sgpo_amd import runtime as rt
    from sgpo_amd import weights as W
    rt.init('cuda:0')
    lib = L.load()
    read_pos = list(range(119))
    vf = rt.MlpValueFunction(W.mlp_synthetic_state(seed=3, n_residues=119), read_pos)
    B = 8192                                   # 8192 * 740 B = 6 MB > 4 MiB
    assert lib.sgpo_mlp_workspace_bytes(vf.handle, B) > lib.sgpo_guidance_noalloc_max_bytes()
    x = torch.zeros(B, 119, dtype=torch.int64, device='cuda')
    out = torch.zeros(B, 119, K, dtype=torch.float32, device='cuda')
    rc = lib.sgpo_guidance_ratio_mlp_absorbing_noalloc(vf.handle, L.ptr(x), 5.0, B, 119, K, MASK, L.ptr(out), L.stream_ptr())
    assert rc != 0
    msg = lib.sgpo_last_error().decode()
    assert 'scratch' in msg and str(4 << 20) in msg
    torch.cuda.synchronize()
    assert float(out.abs().max()) == 0.0       # nothing was launched
