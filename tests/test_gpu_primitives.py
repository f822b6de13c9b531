"""The scan kernels on their own (csrc/primitives.hip): one-wave tiles without an LDS allocation for arrays up to 16 k elements, the
decoupled look-back scan above that -- against numpy's cumsum at the sizes where a tile, a row or the launch path changes, in place
and out of place, on aligned and unaligned arrays (the unaligned ones take the 256-thread kernels), the pair form, and many launches
back to back on one stream (the look-back state is reused with an epoch).  And the radix sort (round 6: tiles ordered in LDS) through
gpcc_sort_zyx: stable, equal to np.lexsort at the sizes where its path or tiling changes, on keys whose upper digits are constant."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SIZES = [1, 3, 63, 64, 65, 255, 256, 257, 1023, 1024, 4095, 4096, 4097, 8191, 16384, 16385, 20000, 65536, 65537, 262144 + 5, 1_000_003, 4_200_001]


def _scan(x, off_in=0, off_out=0, inplace=False, with_total=True):
    from gauspcc_amd import _lib, runtime

    dev = torch.device("cuda", 0)
    ctx = runtime.context(dev)
    n = len(x)
    buf_in = torch.zeros(n + 8, dtype=torch.int32, device=dev)
    a = buf_in[off_in: off_in + n]
    a.copy_(torch.from_numpy(x.astype(np.int32)))
    out = a if inplace else torch.full((n + 8,), -1, dtype=torch.int32, device=dev)[off_out: off_out + n]
    total = torch.full((1,), -1, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(_lib.lib().gpcc_debug_exclusive_scan(ctx, a.data_ptr(), out.data_ptr(), None, None, n, total.data_ptr() if with_total else None, st))
    torch.cuda.synchronize()
    return out.cpu().numpy().astype(np.uint32), int(total.cpu().numpy().astype(np.uint32)[0])


@pytest.mark.parametrize("n", SIZES)
def test_exclusive_scan_matches_cumsum(n):
    rng = np.random.RandomState(n % 9973)
    x = rng.randint(0, 9, size=n).astype(np.uint32)
    ref = np.concatenate([[0], np.cumsum(x[:-1], dtype=np.uint64)]).astype(np.uint32)
    for off_in, off_out, inplace in ((0, 0, False), (0, 0, True), (1, 0, False), (0, 3, False)):
        if n > 1_000_003 and (off_in or off_out):
            continue
        out, total = _scan(x, off_in, off_out, inplace)
        assert np.array_equal(out, ref), (n, off_in, off_out, inplace, int(np.argmax(out != ref)))
        assert total == int(x.sum(dtype=np.uint64) & 0xFFFFFFFF)


def test_scan_wraps_modulo_2_32_and_pair_form():
    from gauspcc_amd import _lib, runtime

    dev = torch.device("cuda", 0)
    ctx = runtime.context(dev)
    st = torch.cuda.current_stream().cuda_stream
    for n in (100, 5000, 16384, 70000):
        rng = np.random.RandomState(n)
        x0 = rng.randint(0, 2 ** 31, size=n, dtype=np.int64).astype(np.uint32)
        x1 = rng.randint(0, 4, size=n).astype(np.uint32)
        a0 = torch.from_numpy(x0.view(np.int32)).to(dev); a1 = torch.from_numpy(x1.view(np.int32)).to(dev)
        o0 = torch.empty_like(a0); o1 = torch.empty_like(a1)
        _lib.check(_lib.lib().gpcc_debug_exclusive_scan(ctx, a0.data_ptr(), o0.data_ptr(), a1.data_ptr(), o1.data_ptr(), n, None, st))
        torch.cuda.synchronize()
        for x, o in ((x0, o0), (x1, o1)):
            ref = (np.concatenate([[0], np.cumsum(x[:-1], dtype=np.uint64)]).astype(np.uint64) & np.uint64(0xFFFFFFFF)).astype(np.uint32)     # modulo 2^32
            assert np.array_equal(o.cpu().numpy().view(np.uint32), ref), n


def test_many_lookback_scans_back_to_back_reuse_their_state():
    rng = np.random.RandomState(3)
    for it in range(40):
        n = int(rng.randint(16385, 600000))
        x = rng.randint(0, 3, size=n).astype(np.uint32)
        out, total = _scan(x)
        assert total == int(x.sum()) and out[-1] == total - int(x[-1]) and out[n // 2] == int(x[: n // 2].sum()), (it, n)


def test_large_scans_on_three_streams_at_once():
    """Launches of 1 024 tiles and more draw their tile ids from eight ticket counters (csrc/primitives.hip: LB_SHARD_MIN_TILES) -- a tile may then meet
    a predecessor whose workgroup has not started yet.  Three streams of one context scan 4.3 M / 5 M / 6.1 M elements (1 050 - 1 490 tiles, every one
    with its own scan state) twelve times each, interleaved with small single-counter scans on the same streams, all in flight together; every
    result against numpy."""
    from gauspcc_amd import _lib, runtime

    dev = torch.device("cuda", 0)
    ctx = runtime.context(dev)
    L = _lib.lib()
    rng = np.random.RandomState(11)
    sizes = [4_300_001, 5_000_000, 6_100_003]
    streams = [torch.cuda.Stream(device=dev) for _ in sizes]
    xs = [rng.randint(0, 4, size=n).astype(np.uint32) for n in sizes]
    refs = [(np.concatenate([np.zeros(1, np.uint64), np.cumsum(x[:-1], dtype=np.uint64)]) & np.uint64(0xFFFFFFFF)).astype(np.uint32) for x in xs]
    small = rng.randint(0, 4, size=70_000).astype(np.uint32)
    small_ref = np.concatenate([np.zeros(1, np.uint64), np.cumsum(small[:-1], dtype=np.uint64)]).astype(np.uint32)
    ins = [torch.from_numpy(x.view(np.int32)).to(dev) for x in xs]
    sm_in = torch.from_numpy(small.view(np.int32)).to(dev)
    outs = [[torch.empty_like(i) for _ in range(12)] for i in ins]
    sm_outs = [[torch.empty_like(sm_in) for _ in range(12)] for _ in sizes]
    totals = [torch.zeros(12, dtype=torch.int32, device=dev) for _ in sizes]
    torch.cuda.synchronize()
    for rep in range(12):
        for k, st in enumerate(streams):
            _lib.check(L.gpcc_debug_exclusive_scan(ctx, ins[k].data_ptr(), outs[k][rep].data_ptr(), None, None, sizes[k], totals[k][rep:].data_ptr(), st.cuda_stream))
            _lib.check(L.gpcc_debug_exclusive_scan(ctx, sm_in.data_ptr(), sm_outs[k][rep].data_ptr(), None, None, small.size, None, st.cuda_stream))
    torch.cuda.synchronize()
    for k in range(len(sizes)):
        assert (totals[k].cpu().numpy().view(np.uint32) == np.uint32(int(xs[k].sum(dtype=np.uint64)) & 0xFFFFFFFF)).all(), k
        for rep in range(12):
            assert np.array_equal(outs[k][rep].cpu().numpy().view(np.uint32), refs[k]), (k, rep)
            assert np.array_equal(sm_outs[k][rep].cpu().numpy().view(np.uint32), small_ref), (k, rep)


# ---------------------------------------------------------------- radix sort (through gpcc_sort_zyx: 63-bit keys with a payload, eight 8-bit passes)
SORT_SIZES = [1, 63, 64, 1023, 1024, 1025, 4095, 4096, 4097, 8193, 12289, 300_001]


def _cloud(kind, n, rng):
    if kind == "random":
        return rng.randint(-(1 << 20), 1 << 20, (n, 3)).astype(np.int32)
    if kind == "x_only":        # y, z constant: the upper six passes see ONE digit per wave (k_radix_hist's one-add path), the payload order must survive them
        p = np.zeros((n, 3), np.int32)
        p[:, 0] = rng.randint(-30000, 30000, n)
        p[:, 1] = 77; p[:, 2] = -5
        return p
    if kind == "dups":          # a few distinct points, each many times: stability (equal keys keep their input order) is the whole result
        base = rng.randint(-100, 100, (7, 3)).astype(np.int32)
        return base[rng.randint(0, 7, n)]
    return np.full((n, 3), 12345, np.int32)     # "constant"


@pytest.mark.parametrize("kind", ["random", "x_only", "dups", "constant"])
@pytest.mark.parametrize("n", SORT_SIZES)
def test_radix_sort_is_stable_and_matches_lexsort(kind, n):
    """csrc/primitives.hip: k_radix_small (n <= 1024, one wave), k_radix_hist / k_radix_scatter (4 096-key tiles ordered in LDS) at the sizes where
    the path, a tile or a wave's share of a tile changes; np.lexsort is stable, so equality of the permutations checks stability too."""
    from tests import gpu_helpers as gh

    rng = np.random.RandomState(n % 9973 + len(kind))
    p = _cloud(kind, n, rng)
    perm = gh.sort_zyx(p)
    assert np.array_equal(perm, np.lexsort((p[:, 0], p[:, 1], p[:, 2])))
