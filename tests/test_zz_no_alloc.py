"""rst_layernorm_bwd_noalloc (partial sums in the library's own per-device array, nothing allocated) against
rst_layernorm_bwd (caller workspace), through the C ABI: two calls in a row, the second on a smaller input, so the
array still holds the first call's partial sums when the second one runs."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _both(lib, B, C, H, W, with_bias, seed):
    from superresolution_amd._lib import check
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, C, H, W, generator=g).cuda()
    dy = torch.randn(B, C, H, W, generator=g).cuda()
    w = (torch.rand(C, generator=g) + 0.5).cuda()
    N = H * W
    out = []
    for noalloc in (False, True):
        dx = torch.full_like(x, float("nan"))
        dw = torch.full((C,), float("nan"), device="cuda")
        db = torch.full((C,), float("nan"), device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        if noalloc:
            def call():
                check(lib.rst_layernorm_bwd_noalloc(x.data_ptr(), w.data_ptr(), int(with_bias), dy.data_ptr(), dx.data_ptr(), 0,
                                                    dw.data_ptr(), db.data_ptr() if with_bias else None, B, C, N, st),
                      "rst_layernorm_bwd_noalloc")
            call()                                    # (the first call on a device looks the array's address up)
            torch.cuda.synchronize()
            # free device memory as the driver sees it: torch's own counter would not notice a hipMalloc of the library's
            free_before, reserved_before = torch.cuda.mem_get_info()[0], torch.cuda.memory_reserved()
            call()
            torch.cuda.synchronize()
            assert torch.cuda.memory_reserved() == reserved_before and torch.cuda.mem_get_info()[0] == free_before
        else:
            nb = lib.rst_layernorm_bwd_workspace_bytes(B, C, N)
            ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
            check(lib.rst_layernorm_bwd(x.data_ptr(), w.data_ptr(), int(with_bias), dy.data_ptr(), dx.data_ptr(), 0,
                                        dw.data_ptr(), db.data_ptr() if with_bias else None, B, C, N, ws.data_ptr(), nb, st),
                  "rst_layernorm_bwd")
        torch.cuda.synchronize()
        out.append((dx, dw, db if with_bias else None))
    return out


def test_noalloc_equals_workspace_form_twice():
    from superresolution_amd import _lib
    lib = _lib.load()
    # first call: many blocks (every partial row of the array gets written); second: a smaller input, fewer blocks
    for (B, C, H, W, wb, seed) in [(4, 96, 64, 96, True, 1), (1, 48, 16, 24, False, 2)]:
        (dx0, dw0, db0), (dx1, dw1, db1) = _both(lib, B, C, H, W, wb, seed)
        assert torch.isfinite(dx1).all() and torch.isfinite(dw1).all()
        assert torch.equal(dx0, dx1)
        assert torch.equal(dw0, dw1)          # same kernels, same fixed summation order: bit-identical
        if wb:
            assert torch.equal(db0, db1)


def test_noalloc_rejects_beyond_the_built_in_scratch():
    from superresolution_amd import _lib
    lib = _lib.load()
    t = torch.zeros(1, 512, 8, 8, device="cuda")
    w = torch.ones(512, device="cuda")
    rc = lib.rst_layernorm_bwd_noalloc(t.data_ptr(), w.data_ptr(), 0, t.data_ptr(), t.data_ptr(), 0, w.data_ptr(), None,
                                       1, 512, 64, None)
    assert rc == 1 and b"384" in lib.rst_last_error()
