"""CPU tests: the C-ABI library loads and exports every symbol include/charon_hip.h declares (no compute calls)."""
import ctypes
import os
import re

from tests import util


def test_library_exports_every_declared_symbol():
    import charon_amd.api as api
    header = open(os.path.join(util.ROOT, "include", "charon_hip.h")).read()
    declared = set(re.findall(r"\b(chn_[a-z_0-9]+)\s*\(", header))
    declared -= {"chn_index", "chn_stream"}
    assert declared == set(api.EXPORTS), declared ^ set(api.EXPORTS)
    for name in declared:
        assert getattr(api.lib(), name) is not None
    assert "gfx950" in api.version()


def test_struct_layout_matches_header():
    import charon_amd.api as api
    # sizes computed by hand from the header's field list (natural alignment)
    assert ctypes.sizeof(api.IndexDesc) == 336
    assert ctypes.sizeof(api.Model) == 88
    assert ctypes.sizeof(api.StreamCfg) == 24
    assert ctypes.sizeof(api.Batch) == 88
    assert ctypes.sizeof(api.Result) == 64


def test_default_model_is_reference_defaults():
    import charon_amd.api as api
    m = api.default_model(2, 1)
    assert (m.num_categories, m.host_index, m.paired, m.min_length, m.confidence_threshold) == (2, 1, 0, 140, 7)
    assert abs(m.h_pos - 0.1) < 1e-7 and abs(m.h_neg - 0.001) < 1e-9 and m.err_rate == 300.0
    pos = [m.pos_data[0][i] for i in range(m.pos_n[0])]
    assert len(pos) == 500 and pos == sorted(pos)
    assert api.default_model(2, 0, paired=True).min_length == 80


def test_packer_roundtrip():
    from charon_amd import pack
    seqs = [b"ACGTNNACGT" * 7, b"", b"acgtRYacgt", b"T" * 64, b"G" * 65]
    p = pack.pack_reads(seqs)
    assert all(int(o) % 64 == 0 for o in p["seg1_offset"])
    back = pack.unpack_reads(p["bases2"], p["seg1_offset"], p["seg1_length"], p["nmask"])
    assert back == [b"ACGTNNACGT" * 7, b"", b"ACGTNNACGT", b"T" * 64, b"G" * 65]
