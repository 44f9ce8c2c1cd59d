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
    assert ctypes.sizeof(api.Model) == 104  # + dist, pos_params, neg_params (round 2)
    assert ctypes.sizeof(api.StreamCfg) == 24
    assert ctypes.sizeof(api.Batch) == 96  # + gzip_tallies (round 2)
    assert ctypes.sizeof(api.Result) == 80  # + gzip_tallies, gzip_sizes (round 2)


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


def test_no_cpu_fallback_when_the_library_is_missing(tmp_path):
    """the product path must fail loudly without libcharon_hip.so: importing the binding from a tree that has no built library
    raises, and nothing under charon_amd/ imports the oracle"""
    import shutil
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = tmp_path / "charon_amd"
    shutil.copytree(os.path.join(root, "charon_amd"), pkg, ignore=shutil.ignore_patterns("*.so", "bin", "csrc", "__pycache__"))
    p = subprocess.run([sys.executable, "-c", "import charon_amd.api"], cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode != 0 and b"not built" in p.stderr
    for dirpath, _, files in os.walk(os.path.join(root, "charon_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".inc", ".h")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "import oracle" not in text and "from oracle" not in text and "liboracle" not in text and "charon_oracle" not in text, (dirpath, f)
