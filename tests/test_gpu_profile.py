"""The measurement hooks of the C ABI (include/gauspcc.h: gpcc_profile_enable / _get / _stages), which bench.py's
`roofline` object is built from: HIP-event brackets around the conv launches and around the HBM-bound stages."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X: the HIP path has no fallback")
    from gauspcc_amd import _lib, runtime
    from gauspcc_amd.synth import synthetic_cloud, synthetic_state_dict
    from tests import gpu_helpers as gh

    model = runtime.Model(synthetic_state_dict(32, 3), 32, 3, 0)
    return gh, _lib, runtime.context(gh.dev()), model, synthetic_cloud(30_000, seed=5)


def _get(_lib, ctx):
    """(ms, convolutions, pair jobs) over BOTH families: the k_sparse_conv launches and the decoder's persistent small-level
    launches (csrc/fused.hpp), which count as the convolutions they contain"""
    p = _lib.Profile()
    _lib.check(_lib.lib().gpcc_profile_get(ctx, C.byref(p)))
    return p.conv_ms + p.fused_ms, p.conv_launches + p.fused_launches, p.conv_pair_jobs + p.fused_pair_jobs


def test_conv_brackets_count_every_launch_and_pair(env):
    """conv_pair_jobs is what `roofline.achieved` multiplies by 2 * C * C: it must equal the pairs the two calls report
    (gpcc_stats.conv_pairs = sum over levels of pairs x convolutions on the level), and the launch count is the same from
    call to call."""
    gh, _lib, ctx, model, pts = env
    L = _lib.lib()
    gh.encode(model, pts, 10)   # warm-up: workspace growth
    _lib.check(L.gpcc_profile_enable(ctx, 1))
    data, st_e = gh.encode(model, pts, 10)
    ms_e, n_e, pj_e = _get(_lib, ctx)
    dec, _, st_d = gh.decode(model, data)
    ms, n, pj = _get(_lib, ctx)
    assert dec.shape == pts.shape
    assert n_e == 12 and ms_e > 0.0                     # two trunks of five + the two batched stage launches
    assert pj_e == st_e.conv_pairs
    assert n - n_e == 18 * (st_d.num_levels - 1)        # a decode: 5 (parent trunk) + 5 + 8 per coded level
    assert pj - pj_e == st_d.conv_pairs == st_e.conv_pairs
    assert ms > ms_e
    # pause: nothing more is recorded, what was collected stays readable
    _lib.check(L.gpcc_profile_enable(ctx, 3))
    gh.encode(model, pts, 10)
    assert _get(_lib, ctx) == (ms, n, pj)
    # off: the accumulators are reset
    _lib.check(L.gpcc_profile_enable(ctx, 0))
    assert _get(_lib, ctx) == (0.0, 0, 0)


def test_stage_brackets(env):
    gh, _lib, ctx, model, pts = env
    L = _lib.lib()
    _lib.check(L.gpcc_profile_enable(ctx, 2))
    data, _ = gh.encode(model, pts, 10)
    gh.decode(model, data)
    arr = (_lib.Stage * 8)()
    ns = C.c_int()
    _lib.check(L.gpcc_profile_stages(ctx, arr, 8, C.byref(ns)))
    _lib.check(L.gpcc_profile_enable(ctx, 0))
    names = [arr[i].name.decode() for i in range(ns.value)]
    assert len(names) == 5 and any("octree" in s for s in names) and any("range coder" in s for s in names)
    for i in range(ns.value):
        assert arr[i].ms > 0.0 and arr[i].bytes > 0.0 and arr[i].brackets > 0, names[i]
    # the algorithmic bytes of the streaming stages scale with the nodes: >= 128 B per coded node for the heads
    heads = [arr[i] for i in range(ns.value) if "heads" in arr[i].name.decode()][0]
    assert heads.bytes > 128.0 * 4 * 30_000


def test_independent_contexts_run_concurrently():
    """Two contexts on two host threads and streams ('independent contexts ... run concurrently on one GPU';
    bench.py's scenes_in_flight pass): every call gives the bytes and the decoded order of the same call made alone."""
    import hashlib
    import threading

    import torch

    from gauspcc_amd import _lib, runtime
    from gauspcc_amd.synth import synthetic_cloud, synthetic_state_dict
    from tests import gpu_helpers as gh

    L = _lib.lib()
    model = runtime.Model(synthetic_state_dict(32, 5), 32, 5, 0)
    clouds = [synthetic_cloud(60_000, seed=11), synthetic_cloud(45_000, seed=12, negative=True)]
    alone = []
    for pts in clouds:
        data, _ = gh.encode(model, pts, 10)
        dec, _, _ = gh.decode(model, data)
        alone.append((hashlib.sha256(data).hexdigest(), hashlib.sha256(dec.tobytes()).hexdigest()))
    xs = [torch.tensor(np.ascontiguousarray(p, dtype=np.int32), device=gh.dev()) for p in clouds]
    ctxs = []
    for _ in clouds:
        h = C.c_void_p()
        _lib.check(L.gpcc_ctx_create(0, C.byref(h)))
        ctxs.append(h)
    streams = [torch.cuda.Stream(device=gh.dev()) for _ in clouds]
    got = [[] for _ in clouds]
    errs = []

    def work(i):
        try:
            for _ in range(4):
                pb, nb, s1 = C.c_void_p(), C.c_int64(), _lib.Stats()
                sp = C.c_void_p(streams[i].cuda_stream)
                _lib.check(L.gpcc_encode(ctxs[i], model.handle, xs[i].data_ptr(), xs[i].shape[0], 10, runtime.f16_bits(1), C.byref(pb), C.byref(nb), C.byref(s1), sp))
                data = C.string_at(pb, nb.value)
                px, nn, pq, s2 = C.c_void_p(), C.c_int64(), C.c_uint16(), _lib.Stats()
                _lib.check(L.gpcc_decode(ctxs[i], model.handle, pb, nb.value, C.byref(px), C.byref(nn), C.byref(pq), C.byref(s2), sp))
                out = torch.empty((nn.value, 3), dtype=torch.int32, device=gh.dev())
                _lib.check(L.gpcc_memcpy_d2d(ctxs[i], out.data_ptr(), px, 12 * nn.value, sp))
                streams[i].synchronize()
                got[i].append((hashlib.sha256(data).hexdigest(), hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()))
        except Exception as e:   # noqa: BLE001 - reported below, on the main thread
            errs.append(repr(e))

    th = [threading.Thread(target=work, args=(i,)) for i in range(len(clouds))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for h in ctxs:
        L.gpcc_ctx_destroy(h)
    assert not errs, errs
    for i in range(len(clouds)):
        assert got[i] == [alone[i]] * 4
