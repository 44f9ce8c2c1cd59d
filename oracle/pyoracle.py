"""ORACLE -- TEST INFRASTRUCTURE ONLY.  ctypes binding of oracle/liboracle.so (the C++14 CPU
restatement, charon_oracle.hpp).  Imported only by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by charon_amd."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
TABLES = os.path.join(ROOT, "charon_amd", "data", "default_kde.txt")


def build():
    subprocess.check_call(["make", "-s", "-C", HERE])


class Thresholds(C.Structure):
    _fields_ = [("min_quality", C.c_float), ("min_length", C.c_uint32), ("min_compression", C.c_float),
                ("confidence_threshold", C.c_uint8), ("confidence_probability_threshold", C.c_float),
                ("host_unique_prop_lo_threshold", C.c_float), ("min_proportion_difference", C.c_float),
                ("min_prob_difference", C.c_float), ("min_hits", C.c_uint8), ("paired", C.c_uint8),
                ("with_gzip", C.c_uint8), ("dist", C.c_uint8)]


def default_thresholds(paired=False, with_gzip=False):
    return Thresholds(15.0, 80 if paired else 140, 0.0, 7, 0.0, 0.05, 0.04, 0.0, 0, 1 if paired else 0,
                      1 if with_gzip else 0, 0)


def classify_thresholds(paired=False, dist="beta", with_gzip=False):
    """`charon classify` defaults (include/classify_arguments.hpp:19-29): call_category for single-end reads too (the `paired`
    field of this struct selects call_category), min_quality 10, min_compression 0.15, confidence 2, min_proportion_diff 0"""
    return Thresholds(10.0, 80 if paired else 140, 0.15, 2, 0.0, 0.05, 0.0, 0.0, 0, 1, 1 if with_gzip else 0,
                      {"kde": 0, "gamma": 1, "beta": 2}[dist])


_lib = None


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        u64p = C.POINTER(C.c_uint64)
        L.orc_kmer_hashes.restype = C.c_uint64
        L.orc_kmer_hashes.argtypes = [C.c_char_p, C.c_uint64, C.c_int, C.c_int, u64p, C.c_uint64]
        L.orc_minimisers.restype = C.c_uint64
        L.orc_minimisers.argtypes = [C.c_char_p, C.c_uint64, C.c_int, C.c_int, C.c_int, u64p, C.c_uint64]
        L.orc_hash_and_fit.restype = C.c_uint64
        L.orc_hash_and_fit.argtypes = [C.c_uint64, C.c_int, C.c_uint64]
        L.orc_bin_size_in_bits.restype = C.c_uint64
        L.orc_bin_size_in_bits.argtypes = [C.c_uint64, C.c_int, C.c_double]
        L.orc_compression_ratio.restype = C.c_float
        L.orc_compression_ratio.argtypes = [C.c_char_p, C.c_uint64]
        L.orc_dexp300.restype = C.c_float
        L.orc_dexp300.argtypes = [C.c_float]
        L.orc_default_model_prob.restype = C.c_double
        L.orc_default_model_prob.argtypes = [C.c_float, C.c_int]
        L.orc_index_new.restype = C.c_void_p
        L.orc_index_new.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_char_p,
                                    C.POINTER(C.c_char_p)]
        L.orc_index_build_from_fasta.restype = C.c_void_p
        L.orc_index_build_from_fasta.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_int,
                                                 C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_uint64]
        L.orc_index_free.argtypes = [C.c_void_p]
        L.orc_index_words.restype = u64p
        L.orc_index_words.argtypes = [C.c_void_p]
        L.orc_index_nwords.restype = C.c_uint64
        L.orc_index_nwords.argtypes = [C.c_void_p]
        L.orc_index_emplace.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
        L.orc_index_emplace_many.argtypes = [C.c_void_p, u64p, C.c_uint64, C.c_uint64]
        L.orc_index_compress.argtypes = [C.c_void_p]
        L.orc_index_use_ef.argtypes = [C.c_void_p, C.c_int]
        L.orc_index_store.argtypes = [C.c_void_p, C.c_char_p]
        L.orc_index_load.restype = C.c_void_p
        L.orc_index_load.argtypes = [C.c_char_p]
        L.orc_index_params.argtypes = [C.c_void_p, u64p]
        L.orc_index_bin_to_cat.argtypes = [C.c_void_p, C.c_char_p]
        L.orc_index_category_name.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int]
        L.orc_index_bulk_contains.argtypes = [C.c_void_p, C.c_uint64, u64p]
        L.orc_sd_get_int.restype = C.c_uint64
        L.orc_sd_get_int.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_sd_has_select.restype = C.c_int
        L.orc_sd_has_select.argtypes = [C.c_void_p]
        L.orc_sd_select_args.restype = C.c_uint64
        L.orc_sd_select_args.argtypes = [C.c_void_p, C.c_int]
        L.orc_sd_select.restype = C.c_uint64
        L.orc_sd_select.argtypes = [C.c_void_p, C.c_int, C.c_uint64]
        L.orc_select_blocks.restype = C.c_int
        L.orc_select_blocks.argtypes = [u64p, C.c_uint64, C.c_char_p]
        L.orc_process_reads.restype = C.c_double
        L.orc_process_reads.argtypes = [C.c_void_p, C.c_char_p, u64p, C.c_uint64, C.c_void_p, C.c_char_p, C.c_float,
                                        C.POINTER(Thresholds), C.c_int] + [C.c_void_p] * 10
        L.orc_dehost_files_dist.restype = C.c_uint64
        L.orc_dehost_files_dist.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_char_p,
                                            C.c_char_p, C.c_uint64]
        L.orc_classify_files.restype = C.c_uint64
        L.orc_classify_files.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.c_uint64]
        L.orc_density.restype = C.c_float
        L.orc_density.argtypes = [C.c_int, C.c_float, C.c_float, C.c_float]
        L.orc_fit.restype = None
        L.orc_fit.argtypes = [C.c_int, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p]
        L.orc_dehost_files.restype = C.c_uint64
        L.orc_dehost_files.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                       C.c_int, C.c_char_p, C.c_uint64]
        L.orc_num_threads.restype = C.c_int
        if L.orc_load_tables(TABLES.encode()) != 0:
            raise RuntimeError("cannot load default KDE tables")
        _lib = L
    return _lib


def _u64p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


def select_blocks(words, nbits, path):
    """the two serialised select_support_mcl blocks (ones, zeros) of a raw bit vector, as store_index appends them"""
    w = np.ascontiguousarray(words, np.uint64)
    assert lib().orc_select_blocks(_u64p(w), nbits, path.encode()) == 0
    return open(path, "rb").read()


def kmer_hashes(seq, k, sigma=5):
    out = np.zeros(max(1, len(seq)), dtype=np.uint64)
    n = lib().orc_kmer_hashes(seq.encode(), len(seq), k, sigma, _u64p(out), len(out))
    return out[:n]


def minimisers(seq, k=19, w=41, sigma=5):
    out = np.zeros(max(1, len(seq)), dtype=np.uint64)
    n = lib().orc_minimisers(seq.encode(), len(seq), k, w, sigma, _u64p(out), len(out))
    return out[:n]


class Index:
    """handle on an oracle::Index"""

    def __init__(self, handle):
        if not handle:
            raise RuntimeError("oracle index handle is NULL")
        self.h = handle
        p = np.zeros(12, dtype=np.uint64)
        lib().orc_index_params(self.h, _u64p(p))
        (self.k, self.w, self.bins, self.tb, self.bin_size, self.hash_shift, self.bin_words, self.hash_funs, self.ncat,
         self.host_index, self.ef_ones, self.ef_wl) = [int(x) for x in p]
        b2c = C.create_string_buffer(max(1, self.bins))
        lib().orc_index_bin_to_cat(self.h, b2c)
        self.bin_to_cat = np.frombuffer(b2c.raw[:self.bins], dtype=np.uint8).copy()
        self.categories = []
        for c in range(self.ncat):
            buf = C.create_string_buffer(256)
            lib().orc_index_category_name(self.h, c, buf, 256)
            self.categories.append(buf.value.decode())

    @classmethod
    def new(cls, bins, bin_size, bin_to_cat, categories, k=19, w=41, nhash=3):
        arr = (C.c_char_p * len(categories))(*[c.encode() for c in categories])
        b2c = bytes(bytearray(int(x) for x in bin_to_cat))
        return cls(lib().orc_index_new(k, w, bins, bin_size, nhash, len(categories), b2c, arr))

    @classmethod
    def from_fasta(cls, files_cats, order, k=19, w=41, force_bin_size=0):
        paths = (C.c_char_p * len(files_cats))(*[p.encode() for p, _ in files_cats])
        cats = (C.c_char_p * len(files_cats))(*[c.encode() for _, c in files_cats])
        od = (C.c_char_p * len(order))(*[c.encode() for c in order])
        return cls(lib().orc_index_build_from_fasta(len(files_cats), paths, cats, len(order), od, k, w, force_bin_size))

    @classmethod
    def load(cls, path):
        return cls(lib().orc_index_load(path.encode()))

    def words(self):
        n = lib().orc_index_nwords(self.h)
        ptr = lib().orc_index_words(self.h)
        return np.ctypeslib.as_array(ptr, shape=(n,))

    def emplace_many(self, values, b):
        values = np.ascontiguousarray(values, dtype=np.uint64)
        lib().orc_index_emplace_many(self.h, _u64p(values), len(values), b)

    def compress(self):
        lib().orc_index_compress(self.h)
        self.__init__(self.h)

    def use_ef(self, on):
        lib().orc_index_use_ef(self.h, 1 if on else 0)

    def store(self, path):
        if lib().orc_index_store(self.h, path.encode()) != 0:
            raise RuntimeError("store failed")

    def bulk_contains(self, value):
        out = np.zeros(self.bin_words, dtype=np.uint64)
        lib().orc_index_bulk_contains(self.h, int(value), _u64p(out))
        return out

    def process_reads(self, seqs, offsets, mate_split=None, quals=None, mq_const=40.0, thr=None, threads=1):
        """seqs: bytes of concatenated ASCII bases, offsets: uint64[n+1].  Returns dict of arrays + seconds."""
        n = len(offsets) - 1
        Cn = self.ncat
        thr = thr or default_thresholds(paired=mate_split is not None)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        out = dict(num_hashes=np.zeros(n, np.uint32), counts=np.zeros((n, Cn), np.uint32), unique=np.zeros((n, Cn), np.uint32),
                   props=np.zeros((n, Cn), np.float32), uprops=np.zeros((n, Cn), np.float32), probs=np.zeros((n, Cn), np.float64),
                   call=np.zeros(n, np.uint8), conf=np.zeros(n, np.uint8), mean_q=np.zeros(n, np.float32),
                   compression=np.zeros(n, np.float32))
        ms = None
        if mate_split is not None:
            ms = np.ascontiguousarray(mate_split, dtype=np.uint32)
        secs = lib().orc_process_reads(self.h, seqs, _u64p(offsets), n, ms.ctypes.data if ms is not None else None, quals,
                                       mq_const, C.byref(thr), threads,
                                       *[out[k].ctypes.data for k in ("num_hashes", "counts", "unique", "props", "uprops", "probs",
                                                                      "call", "conf", "mean_q", "compression")])
        out["seconds"] = secs
        return out

    def dehost_files(self, reads1, reads2="", run_extract=False, chunk_size=100, threads=1, num_reads_to_fit=5000,
                     min_quality=15.0, confidence=7, dist="kde"):
        args = (self.h, reads1.encode(), reads2.encode(), int(run_extract), chunk_size, threads, num_reads_to_fit, min_quality, confidence,
                dist.encode())
        need = lib().orc_dehost_files_dist(*args, None, 0)
        buf = C.create_string_buffer(int(need) + 1)
        lib().orc_dehost_files_dist(*args, buf, need)
        return buf.raw[:need].decode()

    def classify_files(self, reads1, reads2="", run_extract=False, chunk_size=100, threads=1, num_reads_to_fit=5000, dist="beta"):
        args = (self.h, reads1.encode(), reads2.encode(), int(run_extract), chunk_size, threads, num_reads_to_fit, dist.encode())
        need = lib().orc_classify_files(*args, None, 0)
        buf = C.create_string_buffer(int(need) + 1)
        lib().orc_classify_files(*args, buf, need)
        return buf.raw[:need].decode()

    def free(self):
        if self.h:
            lib().orc_index_free(self.h)
            self.h = None


def density(kind, x, p0, p1):
    """stats::dgamma(x, shape p0, scale p1) (kind "gamma") / stats::dbeta(x, alpha p0, beta p1) (kind "beta") as the oracle restates them"""
    return float(lib().orc_density(1 if kind == "gamma" else 2, x, p0, p1))


def fit(kind, data, start, loc_only=False):
    """GammaParams::fit / fit_loc, BetaParams::fit (include/classify_stats.hpp:127-142,171-191) -> 3 floats"""
    d = np.ascontiguousarray(data, np.float32)
    out = np.array(list(start) + [0.0] * (3 - len(start)), np.float32)
    lib().orc_fit(1 if kind == "gamma" else 2, d.ctypes.data, d.size, int(loc_only), out.ctypes.data)
    return out
