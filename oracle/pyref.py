"""ORACLE -- TEST INFRASTRUCTURE ONLY.

Second, independent restatement (pure python ints + numpy float32) of the arithmetic on the
`charon dehost` hot path, written in a deliberately different style from charon_oracle.hpp
(no rolling hashes, every window recomputed from scratch through an explicit deque simulation,
python big-int fastrange, struct-based file codec).  tests/ require it to agree bit-for-bit with
the C++ oracle on the committed fixtures (SURVEY 8(c) pin 2).  Small inputs only -- it is slow.

Reference citations are relative to /root/reference.
"""
import math
import struct
import zlib

import numpy as np

SEED = 0x8F3F73B5CF1C9ADE
IBF_SEEDS = [13572355802537770549, 13043817825332782213, 10650232656628343401, 16499269484942379435,
             4893150838803335377]
M64 = (1 << 64) - 1

RANK5 = {"A": 0, "C": 1, "G": 2, "N": 3, "T": 4}
COMP5 = [4, 2, 1, 3, 0]


def rank5(ch):
    ch = ch.upper()
    if ch == "U":
        ch = "T"
    return RANK5.get(ch, 3)  # every other IUPAC code folds to N (seqan3 dna5)


def kmer_value(digits, sigma):
    v = 0
    for d in digits:
        v = v * sigma + d
    return v


def canonical_values(seq, k, sigma=5):
    """v[i] = min(hash(kmer_i)^seed, hash(revcomp(kmer_i))^seed)   """
    if sigma == 5:
        r = [rank5(c) for c in seq]
        comp = COMP5
    else:
        r = ["ACGT".index(c) for c in seq]
        comp = [3, 2, 1, 0]
    out = []
    for i in range(len(r) - k + 1):
        kmer = r[i:i + k]
        f = kmer_value(kmer, sigma)
        rc = kmer_value([comp[d] for d in reversed(kmer)], sigma)
        out.append(min(f ^ SEED, rc ^ SEED))
    return out


def minimisers(seq, k=19, w=41, sigma=5):
    """seqan3::views::minimiser_hash emission rule, simulated literally with a window list."""
    v = canonical_values(seq, k, sigma)
    if not v:
        return []
    wn = w - k + 1
    win = list(v[:wn])
    out = []

    def recompute():
        # std::ranges::min_element(window, std::less_equal) -> rightmost minimum
        best = 0
        for t in range(1, len(win)):
            if win[t] <= win[best]:
                best = t
        return win[best], best

    mv, off = recompute()
    out.append(mv)
    for x in v[wn:]:
        win.pop(0)
        win.append(x)
        if off == 0:
            mv, off = recompute()
            out.append(mv)
        elif x < mv:
            mv, off = x, len(win) - 1
            out.append(mv)
        else:
            off -= 1
    return out


def clz64(x):
    return 64 - x.bit_length()


def hash_and_fit(x, seed_idx, bin_size):
    """interleaved_bloom_filter::hash_and_fit (fastrange variant), SURVEY A.3"""
    shift = clz64(bin_size)
    x = (x * IBF_SEEDS[seed_idx]) & M64
    x ^= x >> shift
    x = (x * 11400714819323198485) & M64
    return (x * bin_size) >> 64


def bin_size_in_bits(n, h=3, fpr=0.01):
    """src/utils.cpp:75-90"""
    return int(math.ceil(-(n * h) / math.log(1 - math.exp(math.log(fpr) / h))))


class PlainIBF:
    def __init__(self, bins, bin_size, h=3):
        self.bins, self.bin_size, self.h = bins, bin_size, h
        self.words_per_row = (bins + 63) // 64
        self.tb = self.words_per_row * 64
        self.data = np.zeros(bin_size * self.words_per_row, dtype=np.uint64)

    def emplace(self, value, b):
        for i in range(self.h):
            row = hash_and_fit(value, i, self.bin_size)
            self.data[row * self.words_per_row + b // 64] |= np.uint64(1 << (b % 64))

    def bulk_contains(self, value):
        res = [M64] * self.words_per_row
        for i in range(self.h):
            row = hash_and_fit(value, i, self.bin_size)
            for wd in range(self.words_per_row):
                res[wd] &= int(self.data[row * self.words_per_row + wd])
        return res


def count_read(ibf, mins, bin_to_cat, ncat):
    """ReadEntry::get_counts + get_proportions (include/read_entry.hpp:92-150)"""
    rows = [ibf.bulk_contains(m) for m in mins]
    total = [0] * ibf.bins
    for r in rows:
        for b in range(ibf.bins):
            total[b] += (r[b // 64] >> (b % 64)) & 1
    chosen = [255] * ncat
    counts = [0] * ncat
    for b in range(ibf.bins):
        c = bin_to_cat[b]
        if chosen[c] == 255 or total[b] > total[chosen[c]]:
            chosen[c] = b
            counts[c] = total[b]
    unique = [0] * ncat
    for r in rows:
        found = [c for c in range(ncat) if chosen[c] != 255 and (r[chosen[c] // 64] >> (chosen[c] % 64)) & 1]
        if len(found) == 1:
            unique[found[0]] += 1
    n = np.float32(len(mins))
    with np.errstate(divide="ignore", invalid="ignore"):
        props = [np.float32(c) / n for c in counts]
        uprops = [np.float32(u) / n for u in unique]
    return len(mins), counts, unique, props, uprops


def load_tables(path):
    toks = open(path).read().split()
    out, i = {}, 0
    while i < len(toks):
        tag, n = toks[i], int(toks[i + 1])
        out[tag] = np.array([np.float32(float(t)) for t in toks[i + 2:i + 2 + n]], dtype=np.float32)
        i += 2 + n
    return out


def kde_prob(x, dataset_sorted, h):
    """KDEParams::prob (include/classify_stats.hpp:242-252): float accumulate, double exp."""
    x = np.float32(x)
    h = np.float32(h)
    total = np.float32(0)
    for xi in dataset_sorted:
        t = np.float32(np.float32(x - xi) / h)
        kd = math.exp(-(float(t) * float(t)) / 2) / math.sqrt(2 * 3.141592653589793238463)
        total = np.float32(total + np.float32(kd))
    return np.float32(total / np.float32(h * np.float32(len(dataset_sorted))))


def dexp300(x):
    x = np.float32(x)
    if np.isnan(x):
        return np.float32(np.nan)
    if x < 0:
        return np.float32(0)
    # expf evaluated as the correctly rounded value (glibc's expf is correctly rounded here; numpy's SIMD float32
    # exp is not).  log(300) rounded to float first, as `std::log(300.0f)` does.
    arg = np.float32(np.float32(math.log(300.0)) - np.float32(np.float32(300.0) * x))
    return np.float32(math.exp(float(arg)))


def model_prob(x, tables):
    """Model::prob (include/classify_stats.hpp:370-389) for the default KDE model"""
    x = np.float32(x)
    p_err = dexp300(x)
    p_pos = kde_prob(x, np.sort(tables["pos"]), 0.1)
    p_neg = kde_prob(x, np.sort(tables["neg"]), 0.001)
    if x == 1:
        p_pos = np.float32(1)
    total = np.float32(np.float32(p_err + p_pos) + p_neg)
    return float(np.float32(p_pos / total)), float(np.float32(np.float32(p_err + p_neg) / total))


def call_host(unique, uprops, probs, host, mean_q, length, compression, conf_thr=7, min_q=15.0, min_len=140,
              min_comp=0.0, lo=0.05, min_pd=0.04, min_prd=0.0, cpt=0.0):
    """ReadEntry::call_host (include/read_entry.hpp:218-269) -> (call, confidence)"""
    other = 1 - host
    hu, ou = float(uprops[host]), float(uprops[other])
    hp, op = probs[host], probs[other]
    first, second = (other, host) if hu < ou else (host, other)
    raw = (unique[first] - unique[second]) & 0xFFFFFFFF
    conf = 255 if raw > 255 else raw
    thr = conf_thr - 256 if conf_thr >= 128 else conf_thr  # int8 narrowing
    if conf < thr or np.float32(mean_q) < np.float32(min_q) or length < min_len or np.float32(compression) < np.float32(min_comp):
        return 255, conf
    f = lambda v: float(np.float32(v))
    if hu > ou and hu - ou > f(min_pd) and hp > op and hp - op > f(min_prd) and max(hp * conf, float(conf)) >= f(cpt):
        return host, conf
    if hu < f(lo) and hu < ou and ou - hu > f(min_pd) and hp < op and op - hp > f(min_prd) and max(op * conf, float(conf)) >= f(cpt):
        return other, conf
    return 255, conf


def gzip_ratio(seq):
    """get_compression_ratio (src/utils.cpp:114-124) with gzip-hpp's deflate parameters"""
    c = zlib.compressobj(-1, zlib.DEFLATED, 31, 8, zlib.Z_DEFAULT_STRATEGY)
    data = c.compress(seq.encode()) + c.flush()
    return np.float32(len(data) / len(seq))


# ---- index file codec (cereal binary) ------------------------------------------------
def ef_encode(positions, universe):
    m = len(positions)
    hi = lambda x: x.bit_length() - 1 if x else 0
    logm, logn = hi(m) + 1, hi(universe) + 1
    if logm == logn:
        logm -= 1
    wl = logn - logm
    low = 0
    high = 0
    for k, p in enumerate(positions):
        low |= (p & ((1 << wl) - 1)) << (k * wl)
        high |= 1 << ((p >> wl) + k)
    return wl, m * wl, low, m + (1 << logm), high


def _int_vector(width, bits, value):
    nwords = (bits + 63) // 64
    return struct.pack("<BfQQ", width, 1.5, nwords, bits) + value.to_bytes(nwords * 8, "little")


# ---- sdsl select_support_mcl over m_high (the two blocks that close an sd_vector in the file) -----
# Third, independent statement of the layout ([3P-recall] of sdsl-lite v3's select_support_mcl.hpp): a writer that works from
# the list of arg positions, a parser of the stored bytes, and select() as sdsl reads the stored blocks.
def _hi(x):
    return x.bit_length() - 1 if x else 0


def _pack(width, n, vals):
    v = 0
    for i, x in enumerate(vals):
        v |= (x & ((1 << width) - 1)) << (i * width)
    return _int_vector(width, n * width, v)


def bit_array(bits, nbits):
    """python int (bit i = position i) -> numpy array of nbits 0/1 bytes"""
    if isinstance(bits, np.ndarray):
        return bits
    raw = np.frombuffer(int(bits).to_bytes((nbits + 7) // 8 + 1, "little"), np.uint8)
    return np.unpackbits(raw, bitorder="little")[:nbits]


def select_blocks_write(bits, nbits, b):
    """serialised select_support_mcl<b> of the bit vector `bits` (python int, bit i = position i, or a 0/1 array) of nbits bits"""
    pos = [int(x) for x in np.flatnonzero(bit_array(bits, nbits) == b)]
    logn = _hi(((nbits + 63) >> 6) << 6) + 1
    out = struct.pack("<QIII", len(pos), logn, logn * logn, (logn * logn) ** 2)
    if not pos:
        return out
    fast = nbits >= 100000
    sb = (len(pos) + 4095) // 4096
    first, kinds, blocks = [], [], []
    for s_ in range(sb):
        blk = pos[s_ * 4096:(s_ + 1) * 4096]
        tail = fast and len(blk) <= 4032
        # init_fast measures a superblock up to the first arg of the NEXT one (its scan for the last arg counts one arg too many)
        last = pos[(s_ + 1) * 4096] if fast and (s_ + 1) * 4096 < len(pos) else blk[-1]
        if tail or last - blk[0] > (logn * logn) ** 2:
            kinds.append(0)
            first.append(0 if tail else blk[0])
            blocks.append(_pack(_hi(nbits - 1 if tail else last) + 1, 4096, blk))
        else:
            kinds.append(1)
            first.append(blk[0])
            blocks.append(_pack(_hi(last - blk[0]) + 1, 64, [p - blk[0] for p in blk[::64]]))
    out += _pack(logn, sb, first)
    out += _pack(1, sb, kinds) if 0 in kinds else _int_vector(1, 0, 0)
    return out + b"".join(blocks)


def select_blocks_parse(buf, off=0):
    """-> (dict(arg_cnt, logn, superblock, mini_or_long, blocks), offset behind the block); vectors as (width, n, int)"""
    def vec():
        nonlocal off
        width, gf, nwords, bits = struct.unpack_from("<BfQQ", buf, off)
        off += 21
        assert gf == 1.5 and 1 <= width <= 64 and nwords * 64 >= bits and bits % width == 0
        v = int.from_bytes(buf[off:off + nwords * 8], "little")
        off += nwords * 8
        return (width, bits // width, v)

    arg_cnt, logn, logn2, logn4 = struct.unpack_from("<QIII", buf, off)
    off += 20
    d = dict(arg_cnt=arg_cnt, logn=logn, logn2=logn2, logn4=logn4, superblock=None, mini_or_long=None, blocks=[])
    if arg_cnt:
        sb = (arg_cnt + 4095) >> 12
        d["superblock"] = vec()
        d["mini_or_long"] = vec()
        assert d["superblock"][1] == sb and d["mini_or_long"][0] == 1 and d["mini_or_long"][1] in (0, sb)
        d["blocks"] = [vec() for _ in range(sb)]
    return d, off


def _vget(v, i):
    width, n, val = v
    assert i < n
    return (val >> (i * width)) & ((1 << width) - 1)


def select_from_blocks(d, bits, b, i):
    """select_b(i), i from 1, read from the STORED blocks the way select_support_mcl::select does (bits: 0/1 array, see bit_array)"""
    i -= 1
    sb_idx, offset = i >> 12, i & 0xFFF
    mol = d["mini_or_long"]
    if mol[1] and not _vget(mol, sb_idx):  # long superblock: the position verbatim
        return _vget(d["blocks"][sb_idx], offset)
    pos = _vget(d["superblock"], sb_idx) + _vget(d["blocks"][sb_idx], offset >> 6)
    need = offset & 0x3F
    while need:  # scan the bit vector behind the sampled arg
        pos += 1
        if bits[pos] == b:
            need -= 1
    return pos


def write_index(path, k, w, max_fpr, categories, filepath_to_bin, bin_to_category, num_files, records_per_bin,
                hashes_per_bin, ibf):
    s = lambda x: struct.pack("<Q", len(x)) + x.encode()
    out = struct.pack("<BBd", w, k, max_fpr)
    out += struct.pack("<B", ibf.bins) + struct.pack("<Q", len(categories)) + b"".join(s(c) for c in categories)
    out += struct.pack("<Q", len(filepath_to_bin)) + b"".join(s(p) + struct.pack("<B", b) for p, b in filepath_to_bin)
    out += struct.pack("<Q", len(bin_to_category)) + b"".join(struct.pack("<B", b) + s(c) for b, c in bin_to_category.items())
    out += struct.pack("<I", num_files)
    for d in (records_per_bin, hashes_per_bin):
        out += struct.pack("<Q", len(d)) + b"".join(struct.pack("<BQ", b, v) for b, v in d.items())
    out += struct.pack("<QQQQQQ", ibf.bins, ibf.tb, ibf.bin_size, clz64(ibf.bin_size), ibf.words_per_row, ibf.h)
    positions = []
    for wd, val in enumerate(ibf.data):
        val = int(val)
        while val:
            b = (val & -val).bit_length() - 1
            positions.append(wd * 64 + b)
            val &= val - 1
    wl, lowbits, low, highbits, high = ef_encode(positions, ibf.tb * ibf.bin_size)
    out += struct.pack("<QB", ibf.tb * ibf.bin_size, wl) + _int_vector(wl, lowbits, low) + _int_vector(1, highbits, high)
    out += select_blocks_write(high, highbits, 1) + select_blocks_write(high, highbits, 0)  # m_high_1_select, m_high_0_select
    open(path, "wb").write(out)


def read_index(path):
    buf = open(path, "rb").read()
    off = [0]

    def take(fmt):
        v = struct.unpack_from("<" + fmt, buf, off[0])
        off[0] += struct.calcsize("<" + fmt)
        return v if len(v) > 1 else v[0]

    def string():
        n = take("Q")
        v = buf[off[0]:off[0] + n].decode()
        off[0] += n
        return v

    def int_vector():
        width, gf, nwords, bits = take("BfQQ")
        assert gf == 1.5 and 1 <= width <= 64 and nwords * 64 >= bits
        v = int.from_bytes(buf[off[0]:off[0] + nwords * 8], "little")
        off[0] += nwords * 8
        return width, bits, v

    w, k, fpr = take("BBd")
    nb = take("B")
    cats = [string() for _ in range(take("Q"))]
    f2b = [(string(), take("B")) for _ in range(take("Q"))]
    b2c = {}
    for _ in range(take("Q")):
        b = take("B")
        b2c[b] = string()
    nfiles = take("I")
    rpb = dict(take("BQ") for _ in range(take("Q")))
    hpb = dict(take("BQ") for _ in range(take("Q")))
    bins, tb, bsize, shift, words, h = take("QQQQQQ")
    msize, wl = take("QB")
    lw, lbits, low = int_vector()
    hw, hbits, high = int_vector()
    select1 = select0 = None
    if off[0] < len(buf):  # the sd_vector's two select structures (absent in files of an earlier build of this project)
        select1, off[0] = select_blocks_parse(buf, off[0])
        select0, off[0] = select_blocks_parse(buf, off[0])
        assert off[0] == len(buf), "bytes follow m_high_0_select"
    m = lbits // wl if wl else 0
    ibf = PlainIBF(bins, bsize, h)
    z = k_ = 0
    pos = 0
    while k_ < m:
        if (high >> pos) & 1:
            p = (z << wl) | ((low >> (k_ * wl)) & ((1 << wl) - 1))
            ibf.data[p >> 6] |= np.uint64(1 << (p & 63))
            k_ += 1
        else:
            z += 1
        pos += 1
    return dict(w=w, k=k, max_fpr=fpr, num_bins=nb, categories=cats, filepath_to_bin=f2b, bin_to_category=b2c,
                num_files=nfiles, records_per_bin=rpb, hashes_per_bin=hpb, hash_shift=shift, ibf=ibf, msize=msize,
                high=high, high_bits=hbits, select1=select1, select0=select0)
