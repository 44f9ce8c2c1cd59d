#!/usr/bin/env python3
"""Does the bare random-fetch rate of the chip depend on what the GPU did just before (clocks), on the table's size, on the policy?
chn_index_gather_roof on a 39 GB and a 1 GiB stand-in index: cold, after idling, and right after seconds of load."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import charon_amd.api as api

for name, B, S in (("39 GB (B=100, W=2)", 100, 2437500000), ("1 GiB (B=2, W=1)", 2, 1 << 27)):
    b2c = [b % 2 for b in range(B)]
    idx = api.Index(api.make_desc(B, S, b2c, 2, 0, device=0))
    idx.synth_fill(43, 0.02)
    print(name)
    print("  first call          : default %.1f  nt %.1f G fetches/s" % (idx.gather_roof(False) / 1e9, idx.gather_roof(True) / 1e9), flush=True)
    time.sleep(1.0)
    print("  after 1 s of idling : default %.1f  nt %.1f" % (idx.gather_roof(False) / 1e9, idx.gather_roof(True) / 1e9), flush=True)
    t0 = time.time()
    vals = []
    while time.time() - t0 < 2.0:
        vals.append(idx.gather_roof(True) / 1e9)
    print("  2 s back to back    : nt min %.1f  median %.1f  max %.1f  (%d calls)" % (min(vals), sorted(vals)[len(vals) // 2], max(vals), len(vals)), flush=True)
    if os.environ.get("CHN_ROOF_MEMSET"):
        print("  (table overwritten with byte %s before every call above)" % os.environ["CHN_ROOF_MEMSET"])
    t0 = time.time(); os.environ["CHN_ROOF_ITERS"] = "6000"; r = idx.gather_roof(True); dt = time.time() - t0; del os.environ["CHN_ROOF_ITERS"]
    print("  6000 iterations     : nt %.1f by events; the whole call took %.1f ms of wall clock for %.2f G fetches (incl. a 1/8 warm-up launch)" % (r / 1e9, dt * 1e3, 2048 * 256 * 6000 / 1e9), flush=True)
    idx.destroy()
