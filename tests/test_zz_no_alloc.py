"""GPU: the allocation-free entry point uwir_ln2d_affine_noalloc (include/uwir.h) against uwir_ln2d_affine, through the
C ABI: bit-identical outputs on two calls in a row (the second smaller, so stale partials of the first call sit in the
shared scratch), and a clear error beyond the built-in scratch size.  Collected last (test_zz_*): the round's end runs
pytest -x."""
import ctypes as C

import pytest
import torch

from underwater_image_restoration_amd import _lib

pytestmark = pytest.mark.gpu


def _run(lib, noalloc, x, B, n, Cc, gamma, beta, ss):
    dev = x.device
    a, s, mean, pooled = (torch.full((B, Cc), float("nan"), device=dev) for _ in range(4))
    st = torch.cuda.current_stream().cuda_stream
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    if noalloc:
        rc = lib.uwir_ln2d_affine_noalloc(p(x), 0, x.stride(0), B, n, Cc, 1e-5, p(gamma), p(beta), p(ss), ss.stride(0),
                                          p(a), p(s), p(mean), p(pooled), st)
    else:
        nbytes = int(lib.uwir_ln2d_workspace_bytes(B, n, Cc))
        ws = torch.empty(nbytes // 8, dtype=torch.float64, device=dev)
        rc = lib.uwir_ln2d_affine(p(x), 0, x.stride(0), B, n, Cc, 1e-5, p(gamma), p(beta), p(ss), ss.stride(0),
                                  p(a), p(s), p(mean), p(pooled), p(ws), nbytes, st)
    assert rc == 0, lib.uwir_last_error()
    torch.cuda.synchronize()
    return a, s, mean, pooled


def test_noalloc_layernorm_matches_workspace_entry_point():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    lib = _lib.load()
    g = torch.Generator().manual_seed(11)
    # first call: a production shape (level-0 WTB norm at batch 4: 128 row chunks); second call: smaller in B, n and C
    for B, n, Cc, ld in ((4, 16384, 192, 192), (2, 1000, 64, 72)):
        x = (torch.randn(B * n, ld, generator=g) * 3 + 1.5).cuda()
        gamma, beta = torch.randn(Cc, generator=g).cuda(), torch.randn(Cc, generator=g).cuda()
        ss = torch.randn(B, 2 * Cc, generator=g).cuda()
        ref = _run(lib, False, x, B, n, Cc, gamma, beta, ss)
        got = _run(lib, True, x, B, n, Cc, gamma, beta, ss)
        for r, o in zip(ref, got):
            assert bool(torch.isfinite(o).all())
            assert torch.equal(r, o)          # same kernels, same partial layout: bit-identical


def test_noalloc_rejects_shapes_beyond_the_builtin_scratch():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    lib = _lib.load()
    cap = int(lib.uwir_ln2d_noalloc_max_bytes())
    B, n, Cc = 64, 65536, 256
    assert int(lib.uwir_ln2d_workspace_bytes(B, n, Cc)) > cap
    x = torch.zeros(16, Cc, device="cuda")   # never read: the size check comes first
    v = torch.zeros(B, Cc, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    rc = lib.uwir_ln2d_affine_noalloc(p(x), 0, Cc, B, n, Cc, 1e-5, p(v), p(v), None, 0, p(v), p(v), None, None,
                                      torch.cuda.current_stream().cuda_stream)
    assert rc == -1 and b"built-in scratch" in lib.uwir_last_error()
