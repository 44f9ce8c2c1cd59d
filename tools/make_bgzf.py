#!/usr/bin/env python3
"""Write a file as BGZF (blocked gzip, SAM specification 4.1) -- what bgzip does; test / benchmark input only.
usage: python tools/make_bgzf.py in out.gz [level] [processes]"""
import struct
import sys
import zlib
from multiprocessing import Pool

BLOCK = 65280


def member(args):
    c, level = args
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    z = co.compress(c) + co.flush()
    return (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(z) + 8 - 1) + z +
            struct.pack("<II", zlib.crc32(c) & 0xFFFFFFFF, len(c)))


def main():
    src, dst = sys.argv[1], sys.argv[2]
    level = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    procs = int(sys.argv[4]) if len(sys.argv) > 4 else 8

    def chunks():
        with open(src, "rb") as f:
            while True:
                c = f.read(BLOCK)
                if not c:
                    break
                yield (c, level)
    with Pool(procs) as pool, open(dst, "wb") as out:
        for m in pool.imap(member, chunks(), chunksize=64):
            out.write(m)
        out.write(member((b"", level)))


if __name__ == "__main__":
    main()
