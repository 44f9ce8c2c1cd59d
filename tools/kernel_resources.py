#!/usr/bin/env python3
"""Registers, spills, scratch and LDS of every kernel in the gfx950 code object bundled in libcharon_hip.so
(reads the clang offload bundle by hand, then `llvm-readelf --notes`).  Usage: tools/kernel_resources.py [filter]"""
import os
import re
import struct
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
so = os.path.join(ROOT, "charon_amd", "libcharon_hip.so")
data = open(so, "rb").read()
i = data.find(b"__CLANG_OFFLOAD_BUNDLE__")
n = struct.unpack_from("<Q", data, i + 24)[0]
off = i + 32
co = None
for _ in range(n):
    o, sz, ts = struct.unpack_from("<QQQ", data, off)
    off += 24
    trip = data[off:off + ts]
    off += ts
    if b"gfx950" in trip:
        co = data[i + o:i + o + sz]
tmp = "/tmp/charon_k.co"
open(tmp, "wb").write(co)
t = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", tmp], capture_output=True, text=True).stdout
flt = sys.argv[1] if len(sys.argv) > 1 else ""
for blk in t.split("- .agpr_count:")[1:]:
    def f(k):
        m = re.search(r"\.%s:\s+(\S+)" % k, blk)
        return m.group(1) if m else "?"
    name = f("name")
    d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if flt and flt not in d:
        continue
    print("%-72s vgpr %3s sgpr %3s sgpr-spill %3s vgpr-spill %3s scratch %5s lds %6s" % (d[:72], f("vgpr_count"), f("sgpr_count"), f("sgpr_spill_count"),
                                                                                        f("vgpr_spill_count"), f("private_segment_fixed_size"), f("group_segment_fixed_size")))
