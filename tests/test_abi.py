"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every
symbol include/gauspcc.h declares, and the Python shim keeps the reference signatures."""
import inspect
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import ctypes

    from gauspcc_amd import _lib

    header = open(os.path.join(ROOT, "include", "gauspcc.h")).read()
    declared = sorted(set(re.findall(r"GPCC_API\s+[\w\s\*]*?\b(g(?:pcc|sac|sge|sr|shac|snn)_\w+)\s*\(", header)))
    assert len(declared) >= 15
    assert sorted(_lib.EXPORTS) == declared
    so = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(so, name), name
    assert so.gpcc_version() >= 100


def test_signatures_match_reference():
    """Parameter names/defaults of HAC/utils/pcc_utils.py:12,24-31,230-237."""
    from gauspcc_amd import pcc_utils

    sig = inspect.signature(pcc_utils.calculate_morton_order)
    assert list(sig.parameters) == ["x"]
    sig = inspect.signature(pcc_utils.compress_point_cloud)
    pos = [(n, p.default) for n, p in sig.parameters.items() if p.kind is p.POSITIONAL_OR_KEYWORD]
    assert pos == [("xyz_quantized", inspect._empty), ("ckpt_path", inspect._empty), ("output_path", inspect._empty),
                   ("channels", 32), ("kernel_size", 5), ("posQ", 1)]
    sig = inspect.signature(pcc_utils.decompress_point_cloud)
    pos = [(n, p.default) for n, p in sig.parameters.items()]
    assert pos == [("bin_file_path", inspect._empty), ("ckpt_path", inspect._empty), ("output_path", None),
                   ("channels", 32), ("kernel_size", 5), ("is_data_pre_quantized", True)]


def test_no_cpu_fallback():
    """The product path must fail loudly without a GPU (this container has none)."""
    import numpy as np
    import torch

    from gauspcc_amd import pcc_utils

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        pcc_utils.compress_point_cloud(np.zeros((4, 3), np.int32), "synthetic", "/tmp/_x/y.bin")
    with pytest.raises(RuntimeError):
        pcc_utils.calculate_morton_order(torch.zeros((4, 3)))


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "gauspcc_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "liborc" not in txt, f
