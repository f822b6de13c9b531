"""The documents the judge reads must stay navigable: DESIGN.md is the CURRENT design in at most 400 lines (the
round-by-round log lives in HISTORY.md), and every `profiles/...`, `tools/...`, `tests/...` path a document names exists."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _paths(text):
    for m in re.finditer(r"`((?:profiles|tools|tests|gauspcc_amd|oracle|include)/[A-Za-z0-9_./-]+)`", text):
        p = m.group(1).rstrip(".,")
        if "*" in p or p.endswith("/") or "<" in p:
            continue
        yield p


def test_design_is_the_current_design_and_short():
    with open(os.path.join(ROOT, "DESIGN.md")) as f:
        lines = f.read().splitlines()
    assert len(lines) <= 400, len(lines)
    assert os.path.exists(os.path.join(ROOT, "HISTORY.md"))
    text = "\n".join(lines)
    for section in ("## 1. The path and its boundary", "## 2. Oracle and parity", "## 3. Data layout in HBM", "## 4. Kernels and the roofline",
                    "## 8. Multi-GPU", "## 10. Out of scope"):
        assert section in text, section


def test_named_files_exist():
    missing = []
    for doc in ("DESIGN.md", "README.md", "INTEGRATION.md", os.path.join("profiles", "README.md")):
        with open(os.path.join(ROOT, doc)) as f:
            text = f.read()
        if doc.endswith(os.path.join("profiles", "README.md")):
            # the table's first column names files of the directory itself
            for m in re.finditer(r"`(r0[1-6][a-z]?_[A-Za-z0-9_.]+)`", text):
                name = m.group(1)
                if not os.path.exists(os.path.join(ROOT, "profiles", name)) and "*" not in name:
                    missing.append((doc, "profiles/" + name))
        for p in _paths(text):
            if not os.path.exists(os.path.join(ROOT, p)):
                missing.append((doc, p))
    # generated / git-ignored artefacts a document may legitimately name
    allowed = {"gauspcc_amd/libgauspcc.so", "oracle/liborc.so", "tools/ubench/mfma_tail", "tools/ubench/grid_sync", "oracle/_ref"}
    missing = [(d, p) for d, p in missing if p not in allowed and not p.startswith("gauspcc_amd/variants")]
    assert not missing, missing
