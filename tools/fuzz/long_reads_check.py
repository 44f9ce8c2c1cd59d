import sys, numpy as np
sys.path.insert(0, "/root/repo")
from tests import util
from oracle import pyoracle as po
import charon_amd.api as api
from charon_amd import pack
po.build()
r = util.rng(5)
gs = [util.random_seq(r, 4_000_000), util.random_seq(r, 500_000)]
oidx = util.build_oracle_index(po, [[g] for g in gs], [0, 1], ["host", "microbial"])
reads = [util.mutate(r, gs[0][1000:3_500_000], 0.03), gs[1][:150], util.mutate(r, gs[1], 0.1), b"ACGTTGCA" * 300000, b"A" * 1_000_000] + util.sample_reads(r, gs, 40, (100, 3000))
g = util.gpu_index_from_oracle(api, oidx)
p = pack.pack_reads(reads, None)
n = len(reads)
st = api.Stream(g, n, p["n_bases"])
st.set_model(api.default_model(2, oidx.host_index))
st.submit_host(p, np.full(n, 40.0, np.float32), np.zeros(n, np.float32))
gpu = st.wait_host()
seqs, offs, split = util.concat(reads, None)
orc = oidx.process_reads(seqs, offs, mate_split=None, mq_const=40.0, threads=8)
util.assert_parity(gpu, orc)
print("long reads ok", gpu["num_hashes"][:5], gpu["conf"][:5], gpu["call"][:5])
