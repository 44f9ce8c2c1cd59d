#!/usr/bin/env python3
"""Random-fetch rate of the 39 GB stand-in index by wavefronts per CU and fetches in flight per thread (chn_index_gather_roof
diagnostics): what does the probe kernel's shape -- 7 wavefronts per CU, two rounds of 64 x h fetches each -- get from the chip?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import charon_amd.api as api
B, S = 100, 2437500000
idx = api.Index(api.make_desc(B, S, [b % 2 for b in range(B)], 2, 0, device=0))
idx.synth_fill(43, 0.02)
print("blocks/CU (x4 wavefronts)  unroll  LDS/block   in flight per CU   nt G fetches/s")
for bpc, lds in ((8, 0), (4, 0), (2, 0), (1, 0), (2, 70000), (1, 150000)):
    for u in (1, 2, 4, 6, 12):
        os.environ["CHN_ROOF_BLOCKS_PER_CU"] = str(bpc); os.environ["CHN_ROOF_UNROLL"] = str(u); os.environ["CHN_ROOF_LDS"] = str(lds)
        os.environ["CHN_ROOF_ITERS"] = str(1200 if bpc * u >= 8 else 2400)
        r = idx.gather_roof(True) / 1e9
        waves = min(bpc, 160000 // max(lds, 1) if lds else bpc) * 4
        print("%9d %16d %10d %18d %16.1f" % (bpc, u, lds, waves * 64 * u, r), flush=True)
