#!/usr/bin/env python3
"""k_gzip_tally hands a read back to the host when its deflate block would hold lit_bufsize - 1 = 16 383 symbols (zlib flushes the block
there).  DNA reads of up to 61 440 letters never get there, so the rule is checked with a diagnostics build whose limit is 300:

    tools/build_diag.sh GZT_SYMBOL_LIMIT=300u
    python tools/gzt_symbol_limit_check.py full && CHARON_HIP_LIB=tools/diag/libcharon_hip_GZT_SYMBOL_LIMIT=300u.so python tools/gzt_symbol_limit_check.py low
    (then compare /tmp/sym_full.npy and /tmp/sym_low.npy: flagged iff the full build's tallies hold >= 300 symbols; everything else identical)
"""
import os, sys, subprocess, json
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
mode = sys.argv[1]
import charon_amd.api as api
from charon_amd import pack
r = np.random.default_rng(9)
reads = [bytes(r.choice(list(b"ACGTN" if i % 3 == 0 else b"ACGT"), int(r.integers(200, 4000))).astype(np.uint8)) for i in range(600)]
g = api.Index(api.make_desc(2, 1 << 16, [0, 1], 2, 0)); g.synth_fill(43, 0.05)
p = pack.pack_reads(reads)
st = api.Stream(g, len(reads), p["n_bases"]); st.set_model(api.default_model(2, 0))
st.submit_host(p, np.full(len(reads), 40.0, np.float32), None, gzip_tallies=61440, gzip_output=2)
out = st.wait_host()
t = out["gzip_tallies"]
np.save("/tmp/sym_%s.npy" % mode, np.concatenate([t[:, :317].astype(np.int64), out["gzip_sizes"].astype(np.int64)[:, None]], axis=1))
print(mode, "flagged", int((t[:, 316] != 0).sum()), "of", len(reads))
