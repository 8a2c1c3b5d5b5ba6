"""The entry point that allocates nothing: ``hivt_csr_build_static`` (scratch in a static device array
inside the library) must give exactly what ``hivt_csr_build`` (caller-provided workspace) gives, on
two calls in a row, the second on a smaller input, so that what the first call left in the shared
scratch cannot leak into the second."""
import ctypes

import pytest
import torch

from hivt_gan_amd import _lib

pytestmark = pytest.mark.gpu


def _both(key: torch.Tensor, num_rows: int, count=None):
    lib = _lib.load()  # straight through the C ABI (ctypes), no wrapper in between
    e = key.numel()
    dev = key.device
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    cptr = ctypes.c_void_p(count.data_ptr()) if count is not None else None
    outs = []
    for static in (False, True):
        # outputs pre-filled with a pattern so that an entry the call should have written shows up
        rowptr = torch.full((num_rows + 1,), -7, dtype=torch.int32, device=dev)
        perm = torch.full((max(e, 1),), -7, dtype=torch.int32, device=dev)
        if static:
            rc = lib.hivt_csr_build_static(key.data_ptr(), e, cptr, num_rows, rowptr.data_ptr(), perm.data_ptr(),
                                           stream)
        else:
            nbytes = lib.hivt_csr_workspace_bytes(e, num_rows)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            rc = lib.hivt_csr_build(key.data_ptr(), e, cptr, num_rows, rowptr.data_ptr(), perm.data_ptr(), None,
                                    ws.data_ptr(), nbytes, stream)
        assert rc == 0, lib.hivt_last_error().decode()
        outs.append((rowptr, perm))
    torch.cuda.synchronize()
    return outs


def _check(key, num_rows, count=None):
    (rp_a, perm_a), (rp_b, perm_b) = _both(key, num_rows, count)
    assert torch.equal(rp_a, rp_b)
    assert torch.equal(perm_a, perm_b)
    # and both equal the stable sort the CSR order is defined by
    live = key.numel() if count is None else int(count.item())
    ref = torch.sort(key[:live].long(), stable=True).indices.to(torch.int32)
    assert torch.equal(perm_b[:live], ref)
    deg = torch.bincount(key[:live].long(), minlength=num_rows)
    assert torch.equal(rp_b.long(), torch.cat([deg.new_zeros(1), deg.cumsum(0)]))


def test_static_scratch_csr_build_matches_workspace_form_twice():
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(11)
    big = torch.randint(0, 50_000, (700_000,), generator=g, dtype=torch.int32).to(dev)
    _check(big, 50_000)
    # second call, smaller input, rows that the first call left non-zero in the shared scratch
    small = torch.randint(0, 300, (4_000,), generator=g, dtype=torch.int32).to(dev)
    _check(small, 300)
    # fixed-capacity list with a device-side live count, then an empty list
    count = torch.tensor([1234], dtype=torch.int32, device=dev)
    _check(small, 300, count)
    _check(torch.empty(0, dtype=torch.int32, device=dev), 17)


def test_static_scratch_rejects_oversize_inputs():
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    key = torch.zeros(8, dtype=torch.int32, device=dev)
    out = torch.zeros(16, dtype=torch.int32, device=dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    rc = lib.hivt_csr_build_static(key.data_ptr(), (1 << 20) + 1, None, 4, out.data_ptr(), out.data_ptr(), stream)
    assert rc == -1 and b"exceed the static scratch" in lib.hivt_last_error()
    rc = lib.hivt_csr_build_static(key.data_ptr(), 8, None, (1 << 20) + 1, out.data_ptr(), out.data_ptr(), stream)
    assert rc == -1


def test_static_scratch_csr_build_inside_a_recorded_graph():
    """What the entry point exists for: after ONE eager call on the device (which looks the scratch array's address
    up, not a capturable operation) the call records into a HIP graph like any kernel launch; replays on new keys
    in the same buffers give what ``hivt_csr_build`` gives."""
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(5)
    e, rows = 30_000, 900
    key = torch.randint(0, rows, (e,), generator=g, dtype=torch.int32).to(dev)
    rowptr = torch.full((rows + 1,), -7, dtype=torch.int32, device=dev)
    perm = torch.full((e,), -7, dtype=torch.int32, device=dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    assert lib.hivt_csr_build_static(key.data_ptr(), e, None, rows, rowptr.data_ptr(), perm.data_ptr(), stream) == 0
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        s = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        rc = lib.hivt_csr_build_static(key.data_ptr(), e, None, rows, rowptr.data_ptr(), perm.data_ptr(), s)
    assert rc == 0, lib.hivt_last_error().decode()
    for seed in (6, 7):
        fresh = torch.randint(0, rows, (e,), generator=torch.Generator().manual_seed(seed), dtype=torch.int32).to(dev)
        key.copy_(fresh)
        rowptr.fill_(-7)
        perm.fill_(-7)
        graph.replay()
        torch.cuda.synchronize()
        ref = torch.sort(fresh.long(), stable=True).indices.to(torch.int32)
        assert torch.equal(perm, ref)
        deg = torch.bincount(fresh.long(), minlength=rows)
        assert torch.equal(rowptr.long(), torch.cat([deg.new_zeros(1), deg.cumsum(0)]))
