"""Format and numerics regression: the stored bitstreams of tests/golden/containers.npz (written by this implementation's
oracle, tests/golden/make_containers.py) must be reproduced byte for byte -- by the oracle on the CPU and by the HIP path
on the GPU -- and must decode to the stored cloud.  Covers the three layouts (reference layout, chunked with 64- and
up-to-1024-symbol chunks) and both kernel sizes."""
import os

import numpy as np
import pytest

CASES = [(k, cl) for k in (5, 3) for cl in (0, 6, 10)]


def _rows(a):
    return a[np.lexsort((a[:, 0], a[:, 1], a[:, 2]))]


@pytest.fixture(scope="module")
def fixture(golden_dir):
    return np.load(os.path.join(golden_dir, "containers.npz"))


@pytest.mark.parametrize("k,cl", CASES)
def test_oracle_reproduces_stored_stream(orc, fixture, synth_model_k5, synth_model_k3, k, cl):
    model = synth_model_k5 if k == 5 else synth_model_k3
    stored = fixture[f"k{k}_chunk{cl}"].tobytes()
    assert orc.encode(model, fixture["points"], chunk_log2=cl) == stored
    dec, posq = orc.decode(model, stored)
    assert float(posq) == 1.0
    assert np.array_equal(_rows(dec), _rows(fixture["points"]))


def test_stored_headers(fixture):
    """Layout bytes that readers rely on: chunked containers start FF FF | version 2 | chunk_log2; the reference layout
    starts with the f16 posQ = 1.0 (00 3C)."""
    assert fixture["k5_chunk0"][:2].tolist() == [0x00, 0x3C]
    for cl in (6, 10):
        assert fixture[f"k5_chunk{cl}"][:4].tolist() == [0xFF, 0xFF, 2, cl]


def _as_version1(stream: bytes) -> bytes:
    """Version 1 cut every level into 2^chunk_log2-symbol chunks; for chunk_log2 <= 7 version 2 does the same, so the
    chunk_log2 = 6 stream with the version byte set to 1 IS the version-1 stream of the same cloud."""
    assert stream[:4] == bytes([0xFF, 0xFF, 2, 6])
    return stream[:2] + bytes([1]) + stream[3:]


def test_oracle_reads_version1(orc, fixture, synth_model_k5):
    dec, _ = orc.decode(synth_model_k5, _as_version1(fixture["k5_chunk6"].tobytes()))
    assert np.array_equal(_rows(dec), _rows(fixture["points"]))


@pytest.mark.gpu
def test_device_reads_version1(fixture):
    from gauspcc_amd import runtime
    from gauspcc_amd.synth import synthetic_state_dict
    from tests import gpu_helpers as gh

    model = runtime.Model(synthetic_state_dict(32, 5), 32, 5, 0)
    dec, _, _ = gh.decode(model, _as_version1(fixture["k5_chunk6"].tobytes()))
    assert np.array_equal(_rows(dec), _rows(fixture["points"]))


@pytest.mark.gpu
@pytest.mark.parametrize("k,cl", CASES)
def test_device_reproduces_stored_stream(fixture, k, cl):
    from gauspcc_amd import runtime
    from gauspcc_amd.synth import synthetic_state_dict
    from tests import gpu_helpers as gh

    model = runtime.Model(synthetic_state_dict(32, k), 32, k, 0)
    stored = fixture[f"k{k}_chunk{cl}"].tobytes()
    data, _ = gh.encode(model, fixture["points"], cl)
    assert data == stored
    dec, posq, _ = gh.decode(model, stored)
    assert float(posq) == 1.0
    assert np.array_equal(_rows(dec), _rows(fixture["points"]))
