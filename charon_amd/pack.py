"""Host-side packer (numpy): ASCII reads -> the 2-bit + N-mask batch layout of include/charon_hip.h.

Mirrors what seqan3 does to a record's sequence on input (dna5: IUPAC ambiguity codes and N fold to
rank 3, lower case accepted; include/utils.hpp:17-19) and lays segments out at 64-base boundaries.
"""
import numpy as np

_CODE = np.full(256, 4, dtype=np.uint8)  # 4 = N
for ch, v in (("A", 0), ("C", 1), ("G", 2), ("T", 3), ("U", 3)):
    _CODE[ord(ch)] = v
    _CODE[ord(ch.lower())] = v


def pack_reads(seqs, mates=None):
    """seqs: list of bytes/str (mate 1 or single-end); mates: optional list of mate 2.
    Returns dict(bases2, nmask, seg1_offset, seg1_length, seg2_offset, seg2_length, n_bases)."""
    n = len(seqs)
    enc = lambda s: s.encode() if isinstance(s, str) else s
    seqs = [enc(s) for s in seqs]
    if mates is not None:
        mates = [enc(s) for s in mates]
    off1 = np.zeros(n, np.uint64)
    len1 = np.array([len(s) for s in seqs], np.uint32)
    off2 = len2 = None
    if mates is not None:
        off2 = np.zeros(n, np.uint64)
        len2 = np.array([len(s) for s in mates], np.uint32)
    pad = lambda x: (int(x) + 63) // 64 * 64
    cur = 0
    for i in range(n):
        off1[i] = cur
        cur += pad(len1[i])
        if mates is not None:
            off2[i] = cur
            cur += pad(len2[i])
    n_bases = max(cur, 64)
    codes = np.zeros(n_bases, np.uint8)
    for i in range(n):
        codes[int(off1[i]):int(off1[i]) + int(len1[i])] = _CODE[np.frombuffer(seqs[i], np.uint8)]
        if mates is not None:
            codes[int(off2[i]):int(off2[i]) + int(len2[i])] = _CODE[np.frombuffer(mates[i], np.uint8)]
    isn = codes == 4
    c2 = np.where(isn, 0, codes).astype(np.uint32).reshape(-1, 16)
    bases2 = np.zeros(n_bases // 16, np.uint32)
    for j in range(16):
        bases2 |= c2[:, j] << np.uint32(2 * j)
    nmask = None
    if isn.any():
        nb = isn.astype(np.uint32).reshape(-1, 32)
        nmask = np.zeros(n_bases // 32, np.uint32)
        for j in range(32):
            nmask |= nb[:, j] << np.uint32(j)
    return dict(bases2=bases2, nmask=nmask, seg1_offset=off1, seg1_length=len1, seg2_offset=off2, seg2_length=len2,
                n_bases=n_bases)


def unpack_reads(bases2, offsets, lengths, nmask=None):
    """inverse of pack_reads for one segment array (used by tests to hand device-made reads to the oracle)"""
    b = np.asarray(bases2, np.uint32)
    codes = np.zeros(b.size * 16, np.uint8)
    for j in range(16):
        codes[j::16] = (b >> np.uint32(2 * j)) & 3
    letters = np.frombuffer(b"ACGT", np.uint8)[codes]
    if nmask is not None:
        m = np.asarray(nmask, np.uint32)
        isn = np.zeros(m.size * 32, bool)
        for j in range(32):
            isn[j::32] = ((m >> np.uint32(j)) & 1).astype(bool)
        letters = np.where(isn[:letters.size], ord("N"), letters).astype(np.uint8)
    return [letters[int(o):int(o) + int(l)].tobytes() for o, l in zip(offsets, lengths)]
