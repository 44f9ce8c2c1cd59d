#!/usr/bin/env python3
"""Does a submitted batch's GPU work (deflate tallies included) run while the host is busy elsewhere?  Times chn_batch_wait
right after the submit and after a host-side pause as long as the GPU work."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po
from tests import util
from charon_amd import api, pack

r = util.rng(5)
gs = [util.random_seq(r, 200000), util.random_seq(r, 200000)]
oidx = util.build_oracle_index(po, [[g] for g in gs], [0, 1], ["host", "microbial"])
g = util.gpu_index_from_oracle(api, oidx)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
base = util.random_seq(r, 5000 * 64)
reads = [base[(i % 60) * 5000 + (i % 7):][:5000] for i in range(n)]
p = pack.pack_reads(reads)
mq = np.full(n, 40.0, np.float32)
st = api.Stream(g, n, p["n_bases"])
st.set_model(api.default_model(2, 0))
for label, pause in (("warm-up", 0.0), ("wait at once", 0.0), ("wait after 0.15 s", 0.15), ("wait at once", 0.0), ("wait after 0.15 s", 0.15)):
    t0 = time.time(); st.submit_host(p, mq, None, gzip_tallies=16384); t1 = time.time()
    time.sleep(pause)
    t2 = time.time(); out = st.wait_host(); t3 = time.time()
    print("%-18s submit %.1f ms   wait %.1f ms" % (label, (t1 - t0) * 1e3, (t3 - t2) * 1e3), flush=True)
# two in flight
t0 = time.time(); st.submit_host(p, mq, None, gzip_tallies=16384); st.submit_host(p, mq, None, gzip_tallies=16384); t1 = time.time()
time.sleep(0.3)
t2 = time.time(); st.wait_host(); t3 = time.time(); st.wait_host(); t4 = time.time()
print("two in flight: submits %.1f ms, after 0.3 s: first wait %.1f ms, second wait %.1f ms" % ((t1 - t0) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3))
