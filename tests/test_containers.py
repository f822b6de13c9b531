"""Format and numerics regression: the stored bitstreams of tests/golden/containers.npz (written by this implementation's
oracle, tests/golden/make_containers.py) must be reproduced byte for byte -- by the oracle on the CPU and by the HIP path
on the GPU -- and must decode to the stored cloud.  Covers the reference layout, container version 3 (what the encoder
writes: two lanes per byte-counted chunk, Rice-coded count differences) and version 2 (round 2's writer: still read, and still reproduced
by the oracle on request), both kernel sizes."""
import os

import numpy as np
import pytest

CASES = [(k, cl) for k in (5, 3) for cl in (0, 6, 10)]          # stored with the version-2 writer (0: reference layout)
CASES_V3 = [(k, cl) for k in (5, 3) for cl in (6, 11)]
CASES_V4 = CASES_V3                                               # version 4: the same layout, the carry-propagating coder in the lanes


def _rows(a):
    return a[np.lexsort((a[:, 0], a[:, 1], a[:, 2]))]


@pytest.fixture(scope="module")
def fixture(golden_dir):
    return np.load(os.path.join(golden_dir, "containers.npz"))


@pytest.mark.parametrize("k,cl", CASES)
def test_oracle_reproduces_stored_stream(orc, fixture, synth_model_k5, synth_model_k3, k, cl):
    model = synth_model_k5 if k == 5 else synth_model_k3
    stored = fixture[f"k{k}_chunk{cl}"].tobytes()
    orc.set_container_version(2)
    try:
        assert orc.encode(model, fixture["points"], chunk_log2=cl) == stored
    finally:
        orc.set_container_version(4)
    dec, posq = orc.decode(model, stored)
    assert float(posq) == 1.0
    assert np.array_equal(_rows(dec), _rows(fixture["points"]))


@pytest.mark.parametrize("k,cl", CASES_V3)
def test_oracle_reproduces_stored_stream_v3(orc, fixture, synth_model_k5, synth_model_k3, k, cl):
    model = synth_model_k5 if k == 5 else synth_model_k3
    stored = fixture[f"k{k}_chunk{cl}_v3"].tobytes()
    orc.set_container_version(3)
    try:
        assert orc.encode(model, fixture["points"], chunk_log2=cl) == stored
    finally:
        orc.set_container_version(4)
    dec, posq = orc.decode(model, stored)
    assert float(posq) == 1.0
    assert np.array_equal(_rows(dec), _rows(fixture["points"]))


@pytest.mark.parametrize("k,cl", CASES_V4)
def test_oracle_reproduces_stored_stream_v4(orc, fixture, synth_model_k5, synth_model_k3, k, cl):
    """Version 4 = the version-3 layout with the carry-propagating range coder in the lanes (oracle/gpcc_oracle.c: cp_encode_core;
    the reference's coder -- arithmetic_kernel.cu:94-163 -- stays in the reference layout and in versions 1-3)."""
    model = synth_model_k5 if k == 5 else synth_model_k3
    stored = fixture[f"k{k}_chunk{cl}_v4"].tobytes()
    assert orc.encode(model, fixture["points"], chunk_log2=cl) == stored
    dec, posq = orc.decode(model, stored)
    assert float(posq) == 1.0
    assert np.array_equal(_rows(dec), _rows(fixture["points"]))
    # the same tables, lanes and symbols as version 3: sizes differ by the coders' flush bytes only
    v3 = fixture[f"k{k}_chunk{cl}_v3"]
    assert abs(len(stored) - len(v3)) <= 0.004 * len(v3) + 8


def test_stored_headers(fixture):
    """Layout bytes that readers rely on: chunked containers start FF FF | version 2 | chunk_log2; the reference layout
    starts with the f16 posQ = 1.0 (00 3C)."""
    assert fixture["k5_chunk0"][:2].tolist() == [0x00, 0x3C]
    for cl in (6, 10):
        assert fixture[f"k5_chunk{cl}"][:4].tolist() == [0xFF, 0xFF, 2, cl]
    for cl in (6, 11):
        assert fixture[f"k5_chunk{cl}_v3"][:4].tolist() == [0xFF, 0xFF, 3, cl]
        assert fixture[f"k5_chunk{cl}_v4"][:4].tolist() == [0xFF, 0xFF, 4, cl]
    # version 3 is never larger than version 2 at the same lane length (chunk_log2 one higher: two lanes per chunk)
    assert len(fixture["k5_chunk11_v3"]) < len(fixture["k5_chunk10"]) and len(fixture["k3_chunk11_v3"]) < len(fixture["k3_chunk10"])


def _v3_table(counts):
    """The version-3 chunk table, restated: LEB128 first count; with more chunks a byte k and the zigzag differences as Rice
    codes (q = z >> k ones, a zero, k low bits; q >= 16: sixteen ones + z in 32 bits), MSB first, zero padded; k = fewest
    bits, smallest on a tie."""
    out = bytearray()
    v = counts[0]
    while v >= 128:
        out.append((v & 127) | 128)
        v >>= 7
    out.append(v)
    if len(counts) < 2:
        return bytes(out)
    zz = [(d << 1) if d >= 0 else ((-d) << 1) - 1 for d in (counts[i] - counts[i - 1] for i in range(1, len(counts)))]
    cost = [sum((z >> k) + 1 + k if (z >> k) < 16 else 48 for z in zz) for k in range(8)]
    k = cost.index(min(cost))
    out.append(k)
    bits = ""
    for z in zz:
        q = z >> k
        bits += "1" * q + "0" + (format(z & ((1 << k) - 1), f"0{k}b") if k else "") if q < 16 else "1" * 16 + format(z, "032b")
    bits += "0" * (-len(bits) % 8)
    out += int(bits, 2).to_bytes(len(bits) // 8, "big") if bits else b""
    return bytes(out)


def test_v3_table_escape_and_ties():
    # a difference of 0 costs k + 1 bits: k = 0 wins the tie; 3000 -> 12 needs the escape at every k <= 7 (5976 >> 7 = 46)
    assert _v3_table([200, 200, 200]) == bytes([0xC8, 0x01, 0, 0b00000000])
    t = _v3_table([3000, 12])
    assert t[:3] == bytes([0xB8, 0x17, 0]) and t[3:] == bytes([0xFF, 0xFF]) + (5975).to_bytes(4, "big")


def test_v3_table_oracle_matches_the_restatement(orc):
    rng = np.random.RandomState(5)
    cases = [[7], [0, 0], [200, 200, 200], [3000, 12], [5, 2 ** 31 - 1, 0, 77],
             list(rng.randint(200, 240, size=500)) + [17],                       # a stream's chunks and its short last one
             list(rng.randint(0, 5, size=64) * 3000), list(rng.randint(0, 2 ** 20, size=33))]
    for counts in cases:
        counts = [int(c) for c in counts]
        tab = orc.chunk_table(counts)
        assert tab == _v3_table(counts)
        got, used = orc.chunk_table_parse(tab + b"\xAA\xBB", len(counts))
        assert used == len(tab) and got.tolist() == counts
        if len(counts) > 1:
            with pytest.raises(ValueError):
                orc.chunk_table_parse(tab[:-1], len(counts))                     # truncated
            bad = bytearray(tab)
            bad[len(_v3_table(counts[:1]))] = 8                                  # the byte behind the first count
            with pytest.raises(ValueError):
                orc.chunk_table_parse(bytes(bad), len(counts))                   # k > 7


@pytest.mark.parametrize("version", [3, 4])
def test_v3_chunk_table_and_lane_layout(orc, fixture, synth_model_k5, version):
    """Versions 3 and 4 by hand: a stream is its chunk table, then per chunk the forward lane's coder bytes followed by the
    backward lane's coder bytes reversed -- checked against the plain coders (version 3: orc.rc_encode, torchac's; version 4:
    orc.cp_encode, the carry-propagating one) on the traced CDFs / symbols."""
    pts = fixture["points"]
    lane_coder = orc.rc_encode if version == 3 else orc.cp_encode
    orc.set_container_version(version)
    try:
        data = orc.encode(synth_model_k5, pts, chunk_log2=6, trace=True)
    finally:
        orc.set_container_version(4)
    assert data[2] == version
    levels = orc.trace()
    L = data[6]
    pos = 8 + 4 * L + 4
    bn = int.from_bytes(data[pos:pos + 4], "little")
    pos += 4 + 13 * bn
    assert int.from_bytes(data[pos:pos + 2], "little") == 4 * (L - 1)
    pos += 2
    checked = 0
    for lv in levels:
        n = len(lv["sym"][0])
        S = 32                                                        # chunk_log2 6: lanes of 32 symbols
        nl = -(-n // S)
        for s in range(4):
            ln = int.from_bytes(data[pos:pos + 4], "little")
            body = data[pos + 4:pos + 4 + ln]
            pos += 4 + ln
            lanes = [lane_coder(lv["cdf"][s][l * S:(l + 1) * S], lv["sym"][s][l * S:(l + 1) * S]) for l in range(nl)]
            counts, want_pay = [], b""
            for c in range(0, nl, 2):
                counts.append(len(lanes[c]) + (len(lanes[c + 1]) if c + 1 < nl else 0))
                want_pay += lanes[c] + (lanes[c + 1][::-1] if c + 1 < nl else b"")
            assert body == _v3_table(counts) + want_pay
            checked += 1
    assert pos == len(data) and checked == 4 * (L - 1)


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
@pytest.mark.parametrize("k,cl,tag", [(k, cl, "") for k, cl in CASES] + [(k, cl, "_v3") for k, cl in CASES_V3] + [(k, cl, "_v4") for k, cl in CASES_V4])
def test_device_reproduces_stored_stream(fixture, k, cl, tag):
    """The device writes the reference layout, version 3 and version 4 byte for byte; it READS every stored layout (version 2 too)."""
    from gauspcc_amd import runtime
    from gauspcc_amd.synth import synthetic_state_dict
    from tests import gpu_helpers as gh

    model = runtime.Model(synthetic_state_dict(32, k), 32, k, 0)
    stored = fixture[f"k{k}_chunk{cl}{tag}"].tobytes()
    if tag in ("_v3", "_v4") or cl == 0:
        data, _ = gh.encode(model, fixture["points"], cl, version=3 if tag == "_v3" else 4)
        assert data == stored
    dec, posq, _ = gh.decode(model, stored)
    assert float(posq) == 1.0
    assert np.array_equal(_rows(dec), _rows(fixture["points"]))


def test_header_points_applies_the_precheck_bounds_before_anything_is_sized():
    """ADVICE (round 5): the wrapper sizes the output tensor from the header's point count before the library's consistency checks run;
    the same cheap bounds (csrc/container.hpp: container_precheck) therefore apply in Python -- a corrupt 100-byte file must not ask
    for gigabytes."""
    import struct

    from gauspcc_amd.pcc_utils import _header_points

    def hdr(levels, npts, pad=200):
        return b"\xff\xff\x04\x0b" + struct.pack("<H", 0x3C00) + bytes([len(levels), 0]) + b"".join(struct.pack("<I", v) for v in levels) + struct.pack("<I", npts) + bytes(pad)

    good = hdr([8, 40, 200], 900)
    assert _header_points(good[:96], len(good)) == 900
    assert _header_points(hdr([8, 40, 200], 1601)[:96], 300) is None            # more than 8 points per finest node
    assert _header_points(hdr([8, 40, 200], 199)[:96], 300) is None             # fewer points than finest nodes
    assert _header_points(hdr([8, 65, 200], 900)[:96], 300) is None             # a level more than 8x its parent
    assert _header_points(hdr([8, 0, 200], 900)[:96], 300) is None
    big = hdr([60, 480, 3840, 30720, 245760, 1966080, 15728640, 125829120, 1006632960], 2**31 - 2, pad=0)
    assert _header_points(big[:96], len(big)) is None                            # 1.1 G nodes cannot come from a 48-byte file
    assert _header_points(big[:96]) is not None                                  # (without the file size only the tree-shape bounds apply)
    assert _header_points(b"\x03\x00" + bytes(94), 96) is None                   # the reference layout carries no count
