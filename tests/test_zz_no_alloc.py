"""bnpvi_leaf_stats_noalloc (include/bnpvi.h): the M-step data pass with its partial-sum slabs in a
static device array, for callers that cannot allocate.  Through the C ABI, against
bnpvi_leaf_stats on the same inputs: bit-identical statistics and masses, on two calls in a row
(the second with a smaller input, so stale slab contents of the first call must not leak), no
change in the device's free memory across the calls, and a clear error beyond the array's size.
Collected last (file name) so that a failure here cannot hide other tests behind `pytest -x`."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _call(fn, dev, N, want_q, stats, mass):
    from bnp_vae_amd import _lib
    p = lambda t: C.c_void_p(0) if t is None else C.c_void_p(t.data_ptr())
    return fn(dev.ctx, p(dev.Z), dev.Dp, p(dev.phi), dev.tree.Lp, p(dev.vid), N, dev.D, dev.tree.L, dev.V,
              1 if want_q else 0, p(stats), p(mass), dev._stream())


@pytest.mark.parametrize("D,bf,levels,N1,N2,want_q,exact", [
    (64, 8, 2, 999_936, 300_000, False, True),      # BASELINE config 2: LDS-DMA ring kernel, 16.3 MB of slabs
    (64, 8, 3, 200_000, 70_001, False, False),      # config-3 shape: workgroup-wide kernel; the allocating entry
                                                    # point uses 133 MB of slabs, this one fewer, longer chunks
    (32, 5, 2, 60_000, 1_000, True, True),          # config-1 shape with Q: register-streamed kernel
])
def test_leaf_stats_noalloc_matches_allocating_entry_point(D, bf, levels, N1, N2, want_q, exact):
    import torch
    from bench import regular_tree, make_latents
    from bnp_vae_amd import _lib
    from bnp_vae_amd.engine import DeviceVI
    lib = _lib.lib
    parent, is_leaf = regular_tree(bf, levels)
    K, L = len(parent), int(is_leaf.sum())
    dev = DeviceVI(device=0)
    Z, vid, V = make_latents(torch, dev.device, N1, D, L, 0)
    dev.set_latents(Z, vid, V)
    dev.set_tree(parent, is_leaf, np.random.RandomState(13).normal(size=(K, D)), np.ones(K))
    dev.default_gamma()
    dev.estep(mode=_lib.ESTEP_HARD, want_stats=False)
    dev.leaf_stats(want_q=False, want_mass=True)
    dev.params_prior()
    dev.estep(want_stats=False)                                   # soft responsibilities
    W = 1 + D * (2 if want_q else 1)
    torch.cuda.synchronize()
    for N in (N1, N2):                                            # two calls in a row, the second smaller
        Vn = (N + 255) // 256
        want_stats = torch.empty((L, W), dtype=torch.float64, device=dev.device)
        want_mass = torch.empty((V, L), dtype=torch.float64, device=dev.device)
        got_stats = torch.full((L, W), -7.0, dtype=torch.float64, device=dev.device)
        got_mass = torch.full((V, L), -7.0, dtype=torch.float64, device=dev.device)
        dev.V = Vn
        _lib.check(_call(lib.bnpvi_leaf_stats, dev, N, want_q, want_stats, want_mass), "bnpvi_leaf_stats")
        torch.cuda.synchronize()
        free0, _tot = torch.cuda.mem_get_info(dev.device)
        _lib.check(_call(lib.bnpvi_leaf_stats_noalloc, dev, N, want_q, got_stats, got_mass), "bnpvi_leaf_stats_noalloc")
        torch.cuda.synchronize()
        free1, _tot = torch.cuda.mem_get_info(dev.device)
        assert free1 == free0, "device memory changed across the no-allocation call"
        if exact:
            assert torch.equal(got_stats, want_stats)
            assert torch.equal(got_mass[:Vn], want_mass[:Vn])
        else:       # fewer, longer chunks to fit the static array: same sums, fp32 partials grouped differently
            scale = float(want_stats.abs().max())
            assert float((got_stats - want_stats).abs().max()) <= 2e-6 * scale
            np.testing.assert_allclose(got_mass[:Vn].cpu().numpy(), want_mass[:Vn].cpu().numpy(), rtol=2e-6, atol=1e-6)
        ref = dev.phi[:N, :L].to(torch.float64).sum(dim=0)
        np.testing.assert_allclose(got_stats[:, 0].cpu().numpy(), ref.cpu().numpy(), rtol=2e-6, atol=1e-4)
        # without masses
        got2 = torch.full((L, W), -7.0, dtype=torch.float64, device=dev.device)
        _lib.check(_call(lib.bnpvi_leaf_stats_noalloc, dev, N, want_q, got2, None))
        torch.cuda.synchronize()
        assert torch.equal(got2, got_stats)
    dev.V = V
    dev.close()


def test_leaf_stats_noalloc_fresh_context_and_limit():
    """A context that has never allocated a workspace: the call still works (nothing to grow), and a
    launch whose slabs exceed the static array is refused with a message naming both sizes."""
    import torch
    from bnp_vae_amd import _lib
    from bnp_vae_amd.engine import DeviceVI
    lib = _lib.lib
    dev = DeviceVI(device=0)
    N, D, L = 40_000, 64, 64
    g = torch.Generator(device=dev.device)
    g.manual_seed(3)
    Z = torch.randn((N, D), generator=g, device=dev.device)
    vid = (torch.arange(N, device=dev.device) // 100).to(torch.int32)
    V = int(vid[-1]) + 1
    dev.set_latents(Z, vid, V)
    parent = np.concatenate([[-1], np.zeros(L, np.int32)]).astype(np.int32)
    is_leaf = np.concatenate([[0], np.ones(L, np.uint8)]).astype(np.uint8)
    dev.set_tree(parent, is_leaf, np.zeros((L + 1, D)), np.ones(L + 1))
    dev.alloc_phi()
    dev.phi[:, :L] = torch.softmax(torch.randn((N, L), generator=g, device=dev.device), dim=1)
    stats = torch.empty((L, 1 + D), dtype=torch.float64, device=dev.device)
    mass = torch.empty((V, L), dtype=torch.float64, device=dev.device)
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info(dev.device)
    _lib.check(_call(lib.bnpvi_leaf_stats_noalloc, dev, N, False, stats, mass))       # first statistics call of this context
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info(dev.device)
    assert free1 == free0
    phi64 = dev.phi[:, :L].to(torch.float64)
    s1 = phi64.T @ Z.to(torch.float64)
    assert float((stats[:, 1:] - s1).abs().max()) <= 2e-6 * float(s1.abs().max())
    np.testing.assert_allclose(stats[:, 0].cpu().numpy(), phi64.sum(0).cpu().numpy(), rtol=2e-6)
    dev.close()
    # beyond the limit: the config-4 tree (L = 7776, D = 256: 16 MB per slab, so two slabs fit) with more
    # frames than two workgroup rows may sum in fp32 (2 x 8192)
    dev = DeviceVI(device=0)
    N, D, L = 40_000, 256, 7776
    dev.set_latents(torch.zeros((N, D), device=dev.device), torch.zeros(N, dtype=torch.int32, device=dev.device), 1)
    parent = np.concatenate([[-1], np.zeros(L, np.int32)]).astype(np.int32)
    is_leaf = np.concatenate([[0], np.ones(L, np.uint8)]).astype(np.uint8)
    dev.set_tree(parent, is_leaf, np.zeros((L + 1, D)), np.ones(L + 1))
    dev.alloc_phi()
    stats = torch.empty((L, 1 + D), dtype=torch.float64, device=dev.device)
    rc = _call(lib.bnpvi_leaf_stats_noalloc, dev, N, False, stats, None)
    assert rc == 1                                                # BNPVI_EINVAL
    msg = lib.bnpvi_last_error().decode()
    assert "no-allocation" in msg and "bytes" in msg, msg
    _lib.check(_call(lib.bnpvi_leaf_stats, dev, N, False, stats, None))               # the allocating one takes it
    torch.cuda.synchronize()
    dev.close()
