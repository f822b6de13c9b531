"""CPU checks of the synthetic inputs behind bench.py's `low_rate` object (gauspcc_amd/synth.py): the generators are deterministic
and duplicate-free, the committed frequency table is the function's output, and the peaky models reach the rates their docstrings
state under the oracle (estimator: network_ue_4stage_conv.py:100-182)."""
import numpy as np


def test_solid_cloud_is_deterministic_and_duplicate_free():
    from gauspcc_amd.synth import solid_cloud

    a, b = solid_cloud(50_000), solid_cloud(50_000)
    assert a.dtype == np.int32 and a.shape == (50_000, 3)
    assert np.array_equal(a, b)
    assert len(np.unique(a, axis=0)) == 50_000
    assert a.min() >= 0 and a.max() < 4096
    assert not np.array_equal(a, solid_cloud(50_000, seed=5))


def test_stage_symbol_frequencies_match_the_committed_table():
    from gauspcc_amd.synth import PEAKY_FREQ_S1M, STAGE_M, stage_symbol_frequencies, synthetic_cloud

    # the table is the 1 M-point bench cloud's; the 250 k-point cloud of the same generator has the same statistics to ~1e-2
    fr = stage_symbol_frequencies(synthetic_cloud(250_000, seed=1234))
    for s, m in enumerate(STAGE_M):
        assert len(PEAKY_FREQ_S1M[s]) == m and abs(sum(PEAKY_FREQ_S1M[s]) - 1.0) < 2e-3
        assert np.abs(fr[s] - np.array(PEAKY_FREQ_S1M[s])).max() < 0.03
    # occupancy 1 (one child in octant 0) -> symbols (0, 0, 0, 1); occupancy 255 -> (1, 1, 3, 15)
    one = stage_symbol_frequencies(np.array([[2 * i, 0, 0] for i in range(64)] + [[1000, 1000, 1000]]))
    assert one[0][0] > 0.5 and one[3][0] < 1.0


def test_peaky_models_code_at_the_stated_rates(orc):
    from gauspcc_amd.model import tensor_table
    from gauspcc_amd.synth import peaky_state_dict, solid_cloud, stage_symbol_frequencies, synthetic_cloud, synthetic_state_dict

    pts = synthetic_cloud(20_000, seed=77)
    rnd = orc.Model(tensor_table(synthetic_state_dict(32, 5), 32, 5), 32, 5)
    pk = orc.Model(tensor_table(peaky_state_dict(32, 5), 32, 5), 32, 5)
    b_rnd, b_pk = len(orc.encode(rnd, pts, chunk_log2=0)), len(orc.encode(pk, pts, chunk_log2=0))
    assert b_pk < 0.75 * b_rnd                       # ~22 bpp -> ~14.5 bpp: the context-free entropy of the stage symbols
    dec, _ = orc.decode(pk, orc.encode(pk, pts, chunk_log2=11))
    assert len(np.unique(np.concatenate([dec, pts]), axis=0)) == len(pts)
    sol = solid_cloud(20_000)
    sm = orc.Model(tensor_table(peaky_state_dict(32, 5, gain=1.0, freq=stage_symbol_frequencies(sol)), 32, 5), 32, 5)
    assert 8 * len(orc.encode(sm, sol, chunk_log2=0)) / len(sol) < 2.5   # bits per POINT: ~0.2 coded nodes per point
