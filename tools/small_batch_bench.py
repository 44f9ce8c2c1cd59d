#!/usr/bin/env python3
"""BASELINE config 5 at small batch sizes: how close is a whole step (ordering -> minimise+probe -> model+call, two batches in
flight) to the minimise+probe kernel alone?  Paired 2 x 150 b reads vs the 8-category 1 GiB index; per batch size: ms per step
without kernel events (the real chain), and with them (the K1 time the events give).
usage: python tools/small_batch_bench.py [pairs ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import charon_amd.api as api  # noqa: E402


def main():
    sizes = [int(x) for x in sys.argv[1:]] or [8192, 65536, 1 << 20]
    B, S, glen = 8, 1 << 27, 1 << 23
    idx = api.Index(api.make_desc(B, S, list(range(B)), B, 0))
    idx.synth_fill(43, 0.215)
    gen = api.synth_genomes(0, 43, B, glen)
    idx.synth_plant(gen, B, glen, list(range(B)))
    for n in sizes:
        rd = api.synth_reads(0, 42, gen, B, glen, 2 * n, 150, 150, 0.05, 0.10, 40.0)
        off = api.device_download(0, rd.seg1_offset, 2 * n * 8, np.uint64)
        ln = api.device_download(0, rd.seg1_length, 2 * n * 4, np.uint32)
        seg = {}
        for name, arr in (("o1", off[0::2]), ("o2", off[1::2]), ("l1", ln[0::2]), ("l2", ln[1::2])):
            seg[name] = api.device_malloc(0, arr.nbytes)
            api.device_upload(0, seg[name], np.ascontiguousarray(arr))
        res = {}
        for prof in (False, True):
            st = api.Stream(idx, n, rd.n_bases, profile=prof)
            st.set_model(api.default_model(B, 0, paired=True))

            def submit():
                st.submit_device(n, rd.n_bases, rd.bases2, seg["o1"], seg["l1"], rd.mean_quality, rd.compression, seg2_offset=seg["o2"], seg2_length=seg["l2"])
            steps = max(20, min(2000, (1 << 24) // n))
            for rep in range(2):  # first repetition = warm-up
                if prof:
                    for w in range(4):
                        st.profile(w, reset=True)
                st.sync()
                t0 = time.perf_counter()
                submit()
                for i in range(steps):
                    if i + 1 < steps:
                        submit()
                    st.wait_device()
                dt = time.perf_counter() - t0
            res[prof] = dt / steps * 1e3
            if prof:
                k1 = st.profile(0)
                res["k1"] = k1[0] / max(k1[1], 1)
            st.destroy()
        print("%8d pairs/batch: step %.4f ms (no events)  %.4f ms (with events)  minimise+probe kernel %.4f ms  -> step / kernel = %.3f" %
              (n, res[False], res[True], res["k1"], res[False] / res["k1"]), flush=True)
        for p in list(seg.values()) + [rd.bases2, rd.seg1_offset, rd.seg1_length, rd.mean_quality, rd.compression]:
            api.device_free(0, p)


if __name__ == "__main__":
    main()
