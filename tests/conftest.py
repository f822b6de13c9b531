import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A plain `pytest tests` on a box without a GPU skips the gpu-marked tests instead of failing in the product's
    no-CPU-fallback guard (the drivers select with -m gpu / -m "not gpu" anyway)."""
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs an MI355X")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle

    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def synth_model_k5(orc):
    from gauspcc_amd.model import tensor_table
    from gauspcc_amd.synth import synthetic_state_dict

    return orc.Model(tensor_table(synthetic_state_dict(32, 5), 32, 5), 32, 5)


@pytest.fixture(scope="session")
def synth_model_k3(orc):
    from gauspcc_amd.model import tensor_table
    from gauspcc_amd.synth import synthetic_state_dict

    return orc.Model(tensor_table(synthetic_state_dict(32, 3), 32, 3), 32, 3)


@pytest.fixture(scope="session", autouse=True)
def _lds_polluter():
    """Developer knob: GAUSPCC_TEST_POLLUTE=<path to tools/liblds_polluter.so> keeps the LDS of every CU full of a
    non-zero pattern while the GPU tests run (a second stream launching a fill kernel over and over), so that a kernel which
    reads LDS it has not written fails its parity test instead of passing on an idle device's zeros."""
    path = os.environ.get("GAUSPCC_TEST_POLLUTE")
    if not path:
        yield
        return
    import ctypes

    import torch

    torch.cuda.init()
    lib = ctypes.CDLL(os.path.abspath(path))
    lib.pollute_start(0, int(os.environ.get("GAUSPCC_TEST_POLLUTE_PATTERN", "0x00010001"), 16))
    yield
    lib.pollute_stop()
