"""Safety evidence for the host side: everything that reads untrusted bytes on the CPU -- the product's
container / chunk-table / varint parsers (csrc/container.hpp, csrc/rc_format.hpp: the code gpcc_decode runs before it touches the
device), the torchac-compatible host coder and the file writer (csrc/hostcoder.hip), and the oracle's own decoder -- built with
AddressSanitizer + UBSan (-fno-sanitize-recover) and driven by a mutation fuzzer (tools/fuzz_host.cpp).  No GPU involved; GPU-side
ASan is not available on this pool and is not attempted.

The long run (10^6 parser mutants, 2 x 10^4 oracle decodes, 10^5 coder rounds) is kept in profiles/r05_asan_fuzz.txt; this test
runs the same binary at the size the CPU suite can afford.  Readers being hardened: HAC/utils/pcc_utils.py:271-276, kit/op.py:40-48."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_parsers_and_oracle_under_asan_ubsan():
    r = subprocess.run([os.path.join(ROOT, "tools", "asan_host.sh"), "--parse", "200000", "--decode", "2000", "--coder", "20000"], capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "fuzz_host: ok" in r.stdout, tail
    assert "AddressSanitizer" not in tail and "runtime error" not in tail, tail
    # the fuzzer reached both outcomes in every stage: mutants that still parse / decode, and mutants that are refused
    line = [l for l in r.stdout.splitlines() if l.startswith("fuzz_host: ok")][0]
    import re

    m = re.search(r"parsers: (\d+) mutants \((\d+) parsed, (\d+) refused\) \| oracle decode: (\d+) mutants \((\d+) clouds, (\d+) refused\) \| coders: (\d+) rounds", line)
    assert m, line
    n_parse, ok_p, bad_p, n_dec, ok_d, bad_d, n_cod = map(int, m.groups())
    assert n_parse >= 200_000 and ok_p > 1000 and bad_p > 1000 and n_dec >= 2000 and ok_d > 0 and bad_d > 0 and n_cod >= 20_000


def test_fuzz_regression_fixtures(orc, golden_dir, synth_model_k3):
    """What the sanitizer run found, frozen (tests/golden/make_fuzz_regressions.py): a base level that is not in raster order used
    to send the oracle's neighbour search out of bounds; a safe reader orders it and decodes the same cloud.  A base node that
    appears twice is an error."""
    fx = np.load(os.path.join(golden_dir, "fuzz_regressions.npz"))
    good, unsorted, dup = (fx[k].tobytes() for k in ("good", "unsorted_base", "duplicate_base"))
    dec, _ = orc.decode(synth_model_k3, good)
    assert np.array_equal(dec, fx["decoded"])
    dec2, _ = orc.decode(synth_model_k3, unsorted)
    assert np.array_equal(dec2, fx["decoded"])
    with pytest.raises(Exception):
        orc.decode(synth_model_k3, dup)


@pytest.mark.gpu
def test_fuzz_regression_fixtures_on_the_device(golden_dir):
    """The product reader on the same damaged containers: it sorts the base level itself (codec.hip), so an unsorted base level
    decodes to the same cloud and a duplicate base node is a FORMAT error."""
    import torch

    from gauspcc_amd import _lib, runtime
    from gauspcc_amd.pcc_utils import _decode_bytes
    from gauspcc_amd.synth import synthetic_state_dict

    dm = runtime.Model(synthetic_state_dict(32, 3), 32, 3, 0)
    fx = np.load(os.path.join(golden_dir, "fuzz_regressions.npz"))
    dev = torch.device("cuda", 0)
    for name in ("good", "unsorted_base"):
        out, _, _ = _decode_bytes(fx[name].tobytes(), dm, dev)
        assert np.array_equal(out.cpu().numpy(), fx["decoded"]), name
    with pytest.raises(_lib.GpccError, match="duplicate"):
        _decode_bytes(fx["duplicate_base"].tobytes(), dm, dev)
