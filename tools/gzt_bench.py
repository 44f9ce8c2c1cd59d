#!/usr/bin/env python3
"""k_gzip_tally alone: device-resident synthetic reads through chn_batch_submit with deflate tallies asked for, against a tiny
index (the classification kernels are then a few percent of the batch).  Prints the wall time per batch; run it under
`rocprofv3 --kernel-trace --stats` for the kernel's own time.  CHARON_HIP_LIB selects a diagnostics build.

    python tools/gzt_bench.py [n_reads] [read_len | min-max] [batches]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import charon_amd.api as api  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 26667
    lens = sys.argv[2] if len(sys.argv) > 2 else "5000"
    lo, hi = (int(x) for x in lens.split("-")) if "-" in lens else (int(lens), int(lens))
    batches = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    g = api.Index(api.make_desc(2, 1 << 20, [0, 1], 2, 0))
    n_gen, glen = 2, 1 << 20
    gen = api.synth_genomes(0, 43, n_gen, glen)
    g.synth_fill(43, 0.05)
    g.synth_plant(gen, n_gen, glen, [0, 1])
    rd = api.synth_reads(0, 42, gen, n_gen, glen, n, lo, hi, 0.05, 0.1, 40.0)
    st = api.Stream(g, n, rd.n_bases)
    st.set_model(api.default_model(2, 0))
    mq = api.device_malloc(0, n * 4)
    api.device_upload(0, mq, np.full(n, 40.0, np.float32))

    def one(tallies):
        st.submit_device(n, rd.n_bases, rd.bases2, rd.seg1_offset, rd.seg1_length, mq, None, gzip_tallies=tallies, gzip_output=1)
        st.wait_device()

    for t in (0, hi):
        one(t)
        t0 = time.perf_counter()
        for _ in range(batches):
            one(t)
        dt = (time.perf_counter() - t0) / batches
        print("%s: %.2f ms per batch of %d reads of %s letters" % ("with deflate tallies" if t else "classification only", dt * 1e3, n, lens))
    st.destroy()
    g.destroy()


if __name__ == "__main__":
    main()
