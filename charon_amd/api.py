"""ctypes binding of libcharon_hip.so (C ABI: include/charon_hip.h).

There is no fallback path: if the library is missing this module raises at import, so a GPU test can
never silently pass on CPU code.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CHARON_HIP_LIB") or os.path.join(HERE, "libcharon_hip.so")  # the override is for A/B diagnostics builds
if not os.path.exists(LIB_PATH):
    raise ImportError("charon_amd: %s not built -- run `python -c 'import __graft_entry__ as g; g.build()'`" % LIB_PATH)
_L = C.CDLL(LIB_PATH)

MINIMISER_SEED = 0x8F3F73B5CF1C9ADE
STREAM_PROFILE = 1
STREAM_TINY_LOG = 2


class IndexDesc(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("kmer_size", C.c_uint8), ("window_size", C.c_uint8),
                ("hash_funs", C.c_uint8), ("num_categories", C.c_uint8), ("host_index", C.c_uint8), ("reserved0", C.c_uint8 * 3),
                ("minimiser_seed", C.c_uint64), ("bins", C.c_uint64), ("technical_bins", C.c_uint64), ("bin_size", C.c_uint64),
                ("hash_shift", C.c_uint64), ("bin_words", C.c_uint64), ("bin_to_category", C.c_uint8 * 256),
                ("row_begin", C.c_uint64), ("row_end", C.c_uint64)]


class Model(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("num_categories", C.c_uint32), ("pos_data", C.POINTER(C.POINTER(C.c_float))),
                ("pos_n", C.POINTER(C.c_uint32)), ("neg_data", C.POINTER(C.POINTER(C.c_float))), ("neg_n", C.POINTER(C.c_uint32)),
                ("h_pos", C.c_float), ("h_neg", C.c_float), ("err_rate", C.c_float), ("min_quality", C.c_float),
                ("min_length", C.c_uint32), ("min_compression", C.c_float), ("confidence_threshold", C.c_int8),
                ("min_hits", C.c_uint8), ("paired", C.c_uint8), ("host_index", C.c_uint8),
                ("confidence_probability_threshold", C.c_float), ("host_unique_prop_lo_threshold", C.c_float),
                ("min_proportion_difference", C.c_float), ("min_prob_difference", C.c_float),
                ("dist", C.c_uint32), ("pos_params", C.POINTER(C.c_float)), ("neg_params", C.POINTER(C.c_float))]


class StreamCfg(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("flags", C.c_uint32), ("max_reads", C.c_uint64), ("max_bases", C.c_uint64)]


class Batch(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("on_device", C.c_uint32), ("n_reads", C.c_uint64), ("n_bases", C.c_uint64),
                ("bases2", C.c_void_p), ("nmask", C.c_void_p), ("seg1_offset", C.c_void_p), ("seg1_length", C.c_void_p),
                ("seg2_offset", C.c_void_p), ("seg2_length", C.c_void_p), ("mean_quality", C.c_void_p),
                ("compression", C.c_void_p), ("gzip_tallies", C.c_uint32), ("gzip_output", C.c_uint32)]


class Result(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("on_device", C.c_uint32), ("num_hashes", C.c_void_p), ("counts", C.c_void_p),
                ("unique_counts", C.c_void_p), ("probabilities", C.c_void_p), ("call", C.c_void_p), ("confidence", C.c_void_p),
                ("flags", C.c_void_p), ("gzip_tallies", C.c_void_p), ("gzip_sizes", C.c_void_p)]


class SynthReadsOut(C.Structure):
    _fields_ = [("bases2", C.c_void_p), ("seg1_offset", C.c_void_p), ("seg1_length", C.c_void_p), ("mean_quality", C.c_void_p),
                ("compression", C.c_void_p), ("n_bases", C.c_uint64)]


# every symbol include/charon_hip.h declares
EXPORTS = ["chn_index_create", "chn_index_upload_rows", "chn_index_device_words", "chn_index_download_rows",
           "chn_index_get_desc", "chn_index_destroy", "chn_model_default", "chn_stream_create", "chn_stream_destroy",
           "chn_model_set", "chn_batch_submit", "chn_batch_wait", "chn_stream_sync", "chn_classify_counts", "chn_stream_profile",
           "chn_stream_last_batch_bytes", "chn_synth_genomes", "chn_synth_fill_index", "chn_synth_plant", "chn_synth_reads",
           "chn_device_free", "chn_device_download", "chn_device_malloc", "chn_device_upload", "chn_host_alloc", "chn_host_free", "chn_shard_minimise",
           "chn_shard_probe", "chn_shard_finish", "chn_shardx_minimise", "chn_shardx_counts", "chn_shardx_queries", "chn_shardx_serve", "chn_shardx_finish", "chn_minimisers", "chn_index_emplace", "chn_index_decode_ef", "chn_index_bin_popcounts", "chn_index_gather_roof", "chn_last_error", "chn_version"]

_L.chn_last_error.restype = C.c_char_p
_L.chn_version.restype = C.c_char_p
_L.chn_index_create.argtypes = [C.POINTER(IndexDesc), C.POINTER(C.c_void_p)]
_L.chn_index_upload_rows.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
_L.chn_index_download_rows.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
_L.chn_index_device_words.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
_L.chn_index_get_desc.argtypes = [C.c_void_p, C.POINTER(IndexDesc)]
_L.chn_index_destroy.argtypes = [C.c_void_p]
_L.chn_model_default.argtypes = [C.POINTER(Model), C.c_uint32, C.c_uint8, C.c_int]
_L.chn_stream_create.argtypes = [C.c_void_p, C.POINTER(StreamCfg), C.POINTER(C.c_void_p)]
_L.chn_stream_destroy.argtypes = [C.c_void_p]
_L.chn_model_set.argtypes = [C.c_void_p, C.POINTER(Model)]
_L.chn_batch_submit.argtypes = [C.c_void_p, C.POINTER(Batch)]
_L.chn_batch_wait.argtypes = [C.c_void_p, C.POINTER(Result)]
_L.chn_stream_sync.argtypes = [C.c_void_p]
_L.chn_classify_counts.argtypes = [C.c_void_p, C.c_uint64] + [C.c_void_p] * 9
_L.chn_stream_profile.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_int]
_L.chn_stream_last_batch_bytes.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
_L.chn_synth_genomes.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(C.c_void_p)]
_L.chn_synth_fill_index.argtypes = [C.c_void_p, C.c_uint64, C.c_double]
_L.chn_index_gather_roof.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double)]
_L.chn_synth_plant.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_char_p]
_L.chn_synth_reads.argtypes = [C.c_int, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32,
                               C.c_double, C.c_double, C.c_float, C.POINTER(SynthReadsOut)]
_L.chn_device_free.argtypes = [C.c_int, C.c_void_p]
_L.chn_device_malloc.argtypes = [C.c_int, C.c_uint64, C.POINTER(C.c_void_p)]
_L.chn_host_alloc.argtypes = [C.c_uint64, C.POINTER(C.c_void_p)]
_L.chn_host_free.argtypes = [C.c_void_p]
_L.chn_device_upload.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint64]
_L.chn_shard_minimise.argtypes = [C.c_void_p, C.POINTER(Batch), C.POINTER(C.c_uint64)]
_L.chn_shard_probe.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
_L.chn_shard_finish.argtypes = [C.c_void_p, C.c_void_p]
_L.chn_shardx_minimise.argtypes = [C.c_void_p, C.POINTER(Batch)]
_L.chn_shardx_counts.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p]
_L.chn_shardx_queries.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
_L.chn_shardx_serve.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
_L.chn_shardx_finish.argtypes = [C.c_void_p, C.c_void_p]
_L.chn_minimisers.argtypes = [C.c_void_p, C.POINTER(Batch), C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
_L.chn_index_emplace.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32]
_L.chn_index_decode_ef.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint64,
                                   C.POINTER(C.c_uint64)]
_L.chn_index_bin_popcounts.argtypes = [C.c_void_p, C.c_void_p]
_L.chn_device_download.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint64]


class ChnError(RuntimeError):
    pass


def _chk(rc):
    if rc != 0:
        raise ChnError("libcharon_hip error %d: %s" % (rc, _L.chn_last_error().decode()))


def lib():
    return _L


def version():
    return _L.chn_version().decode()


def clz64(x):
    return 64 - int(x).bit_length()


def make_desc(bins, bin_size, bin_to_cat, num_categories, host_index, k=19, w=41, hash_funs=3, device=0,
              row_begin=0, row_end=0):
    d = IndexDesc()
    d.struct_size = C.sizeof(IndexDesc)
    d.device = device
    d.kmer_size, d.window_size, d.hash_funs = k, w, hash_funs
    d.num_categories, d.host_index = num_categories, host_index
    d.minimiser_seed = MINIMISER_SEED
    d.bins = bins
    d.bin_words = (bins + 63) // 64
    d.technical_bins = d.bin_words * 64
    d.bin_size = bin_size
    d.hash_shift = clz64(bin_size)
    for b in range(bins):
        d.bin_to_category[b] = int(bin_to_cat[b])
    d.row_begin, d.row_end = row_begin, row_end
    return d


class Index:
    def __init__(self, desc):
        self.desc = desc
        self.h = C.c_void_p()
        _chk(_L.chn_index_create(C.byref(desc), C.byref(self.h)))

    @property
    def device(self):
        return self.desc.device

    def upload(self, words, row_begin=0):
        words = np.ascontiguousarray(words, dtype=np.uint64)
        n_rows = words.size // self.desc.bin_words
        _chk(_L.chn_index_upload_rows(self.h, row_begin, n_rows, words.ctypes.data))

    def download(self, row_begin=0, n_rows=None):
        n_rows = self.desc.bin_size if n_rows is None else n_rows
        out = np.zeros(n_rows * self.desc.bin_words, np.uint64)
        _chk(_L.chn_index_download_rows(self.h, row_begin, n_rows, out.ctypes.data))
        return out

    def device_words(self):
        p, n = C.c_void_p(), C.c_uint64()
        _chk(_L.chn_index_device_words(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def decode_ef(self, m_size, wl, high, high_bits, low, slice_words=1 << 20):
        """decode a whole sd_vector (numpy uint64 arrays `high` with `high_bits` valid bits, `low` packed wl-bit elements) on the
        device in slices of `slice_words` words of m_high; returns the number of ill-placed bits (0 for a well-formed vector)"""
        high = np.ascontiguousarray(high, np.uint64)
        low = np.ascontiguousarray(low, np.uint64)
        n_high = (high_bits + 63) // 64
        ones_before, bad_total = 0, 0
        for w0 in range(0, n_high, slice_words):
            hs = high[w0:min(n_high, w0 + slice_words)]
            ones = int(sum(bin(int(x)).count("1") for x in hs)) if hs.size < 4096 else int(np.unpackbits(hs.view(np.uint8)).sum())
            if ones == 0:
                continue
            elem0 = ones_before & ~63  # a multiple of 64 elements starts on a word boundary whatever wl is
            lw0 = elem0 * wl // 64
            lw1 = ((ones_before + ones) * wl + 63) // 64
            ls = low[lw0:min(low.size, lw1)]
            bad = C.c_uint64()
            _chk(_L.chn_index_decode_ef(self.h, m_size, wl, hs.ctypes.data, w0 * 64, hs.size, ones_before, ls.ctypes.data if ls.size else None,
                                        elem0, ls.size, C.byref(bad)))
            bad_total += bad.value
            ones_before += ones
        bad = C.c_uint64()  # the closing call: waits for the slices in flight and reports the ill-placed bits they met
        _chk(_L.chn_index_decode_ef(self.h, m_size, wl, high.ctypes.data if high.size else np.zeros(1, np.uint64).ctypes.data, 0, 0, 0, None, 0, 0, C.byref(bad)))
        return bad_total + bad.value

    def bin_popcounts(self):
        out = np.zeros(self.desc.technical_bins, np.uint64)
        _chk(_L.chn_index_bin_popcounts(self.h, out.ctypes.data))
        return out

    def emplace(self, values, bin_index):
        values = np.ascontiguousarray(values, dtype=np.uint64)
        _chk(_L.chn_index_emplace(self.h, values.ctypes.data, values.size, bin_index))

    def gather_roof(self, nt=True):
        """row fetches per second this device sustains for nothing but random probes of this index (measurement aid)"""
        r = C.c_double()
        _chk(_L.chn_index_gather_roof(self.h, 1 if nt else 0, C.byref(r)))
        return r.value

    def synth_fill(self, seed, density):
        _chk(_L.chn_synth_fill_index(self.h, seed, density))

    def synth_plant(self, dev_genomes, n_genomes, genome_len, genome_bin):
        _chk(_L.chn_synth_plant(self.h, dev_genomes, n_genomes, genome_len, bytes(bytearray(int(b) for b in genome_bin))))

    def destroy(self):
        if self.h:
            _L.chn_index_destroy(self.h)
            self.h = None


DIST = {"kde": 0, "gamma": 1, "beta": 2}
# Model defaults of include/classify_stats.hpp:265-268: gamma (shape, loc, scale), beta (alpha, beta, -)
DIST_DEFAULTS = {"gamma": ((25.0, 0.0, 0.02), (10.0, 0.0, 0.005)), "beta": ((6.0, 4.0, 0.0), (6.0, 40.0, 0.0))}


def default_model(num_categories, host_index, paired=False, dist="kde", pos_params=None, neg_params=None, **overrides):
    """paired=True selects call_category (paired dehost, every classify run).  dist gamma / beta: per-category parameter triples
    (default: the reference's) instead of the KDE datasets."""
    m = Model()
    _chk(_L.chn_model_default(C.byref(m), num_categories, host_index, 1 if paired else 0))
    if dist != "kde":
        pp = np.ascontiguousarray(pos_params if pos_params is not None else [DIST_DEFAULTS[dist][0]] * num_categories, np.float32)
        nn = np.ascontiguousarray(neg_params if neg_params is not None else [DIST_DEFAULTS[dist][1]] * num_categories, np.float32)
        m._keep = (pp, nn)
        m.dist = DIST[dist]
        m.pos_params = pp.ctypes.data_as(C.POINTER(C.c_float))
        m.neg_params = nn.ctypes.data_as(C.POINTER(C.c_float))
    for k, v in overrides.items():
        setattr(m, k, v)
    return m


class Stream:
    def __init__(self, index, max_reads, max_bases, profile=False, tiny_log=False, split_bucket=0):
        """split_bucket (testing): length class from which single-end reads are cut over the 64 lanes of a wavefront
        (CHN_STREAM_SPLIT_BUCKET: 0 = default, 32 768 bases; 64 = 1 024 bases; 255 = never)"""
        self.index = index
        cfg = StreamCfg(C.sizeof(StreamCfg), (STREAM_PROFILE if profile else 0) | (STREAM_TINY_LOG if tiny_log else 0) | ((split_bucket & 0xff) << 8),
                        max_reads, max_bases)
        self.h = C.c_void_p()
        _chk(_L.chn_stream_create(index.h, C.byref(cfg), C.byref(self.h)))
        self.C = index.desc.num_categories
        self._fifo = []  # (n_reads, keep-alive host arrays) of the batches in flight

    def set_model(self, model):
        _chk(_L.chn_model_set(self.h, C.byref(model)))

    def submit_host(self, packed, mean_quality=None, compression=None, gzip_tallies=0, gzip_output=0):
        """packed: dict from charon_amd.pack.pack_reads; gzip_tallies: longest read to tally on the device (0 = off);
        gzip_output: 0 tallies, 1 gzip member sizes (tree arithmetic on the device as well), 2 both"""
        b, keep, n = self._host_batch(packed, mean_quality, compression)
        b.gzip_tallies, b.gzip_output = gzip_tallies, gzip_output
        _chk(_L.chn_batch_submit(self.h, C.byref(b)))
        self._fifo.append((n, keep, gzip_tallies, gzip_output))

    def _host_batch(self, packed, mean_quality, compression):
        n = len(packed["seg1_length"])
        b = Batch()
        b.struct_size, b.on_device, b.n_reads, b.n_bases = C.sizeof(Batch), 0, n, packed["n_bases"]
        keep = []

        def ptr(a, dt):
            if a is None:
                return None
            a = np.ascontiguousarray(a, dtype=dt)
            keep.append(a)
            return a.ctypes.data

        b.bases2 = ptr(packed["bases2"], np.uint32)
        b.nmask = ptr(packed.get("nmask"), np.uint32)
        b.seg1_offset = ptr(packed["seg1_offset"], np.uint64)
        b.seg1_length = ptr(packed["seg1_length"], np.uint32)
        b.seg2_offset = ptr(packed.get("seg2_offset"), np.uint64)
        b.seg2_length = ptr(packed.get("seg2_length"), np.uint32)
        b.mean_quality = ptr(mean_quality, np.float32)
        b.compression = ptr(compression, np.float32)
        return b, keep, n

    def minimisers_host(self, packed):
        """all minimisers of the batch (with repeats) as a uint64 array"""
        b, keep, n = self._host_batch(packed, None, None)
        cap = int(packed["n_bases"])
        out = np.zeros(max(cap, 1), np.uint64)
        e = C.c_uint64()
        _chk(_L.chn_minimisers(self.h, C.byref(b), out.ctypes.data, cap, C.byref(e)))
        return out[:e.value]

    # ---- row-sharded mode (see include/charon_hip.h) ----
    def shard_minimise_host(self, packed):
        b, keep, n = self._host_batch(packed, None, None)
        e = C.c_uint64()
        _chk(_L.chn_shard_minimise(self.h, C.byref(b), C.byref(e)))
        self._shard = (n, keep)
        return e.value

    def shard_minimise_device(self, n_reads, n_bases, bases2, seg1_offset, seg1_length, mean_quality=None, compression=None):
        b = Batch()
        b.struct_size, b.on_device, b.n_reads, b.n_bases = C.sizeof(Batch), 1, n_reads, n_bases
        b.bases2, b.seg1_offset, b.seg1_length, b.mean_quality, b.compression = bases2, seg1_offset, seg1_length, mean_quality, compression
        e = C.c_uint64()
        _chk(_L.chn_shard_minimise(self.h, C.byref(b), C.byref(e)))
        self._shard = (n_reads, None)
        return e.value

    def shard_probe(self, shard_index, dev_partial, capacity_words):
        _chk(_L.chn_shard_probe(self.h, shard_index.h, dev_partial, capacity_words))

    def shard_finish(self, dev_partial):
        _chk(_L.chn_shard_finish(self.h, dev_partial))
        self._fifo.append(self._shard)
        self._shard = None

    # ---- row-sharded mode, sparse exchange (see include/charon_hip.h) ----
    def shardx_minimise_host(self, packed, mean_quality=None, compression=None):
        b, keep, n = self._host_batch(packed, mean_quality, compression)
        _chk(_L.chn_shardx_minimise(self.h, C.byref(b)))
        self._shard = (n, keep)

    def shardx_minimise_device(self, n_reads, n_bases, bases2, seg1_offset, seg1_length, mean_quality=None, compression=None):
        b = Batch()
        b.struct_size, b.on_device, b.n_reads, b.n_bases = C.sizeof(Batch), 1, n_reads, n_bases
        b.bases2, b.seg1_offset, b.seg1_length, b.mean_quality, b.compression = bases2, seg1_offset, seg1_length, mean_quality, compression
        _chk(_L.chn_shardx_minimise(self.h, C.byref(b)))
        self._shard = (n_reads, None)

    def shardx_counts(self, row_splits):
        sp = np.ascontiguousarray(row_splits, dtype=np.uint64)
        n_ranks = sp.size - 1
        counts = np.zeros(n_ranks, np.uint64)
        total = C.c_uint64()
        _chk(_L.chn_shardx_counts(self.h, n_ranks, sp.ctypes.data, C.byref(total), counts.ctypes.data))
        return total.value, [int(x) for x in counts]

    def shardx_queries(self, dev_queries, capacity):
        _chk(_L.chn_shardx_queries(self.h, dev_queries, capacity))

    def shardx_serve(self, shard_index, dev_queries_in, n_in, dev_rows_out):
        _chk(_L.chn_shardx_serve(self.h, shard_index.h, dev_queries_in, n_in, dev_rows_out))

    def shardx_finish(self, dev_rows_back):
        _chk(_L.chn_shardx_finish(self.h, dev_rows_back))
        self._fifo.append(self._shard)
        self._shard = None

    def submit_device(self, n_reads, n_bases, bases2, seg1_offset, seg1_length, mean_quality=None, compression=None,
                      nmask=None, seg2_offset=None, seg2_length=None, gzip_tallies=0, gzip_output=0):
        b = Batch()
        b.struct_size, b.on_device, b.n_reads, b.n_bases = C.sizeof(Batch), 1, n_reads, n_bases
        b.bases2, b.nmask, b.seg1_offset, b.seg1_length = bases2, nmask, seg1_offset, seg1_length
        b.seg2_offset, b.seg2_length, b.mean_quality, b.compression = seg2_offset, seg2_length, mean_quality, compression
        b.gzip_tallies, b.gzip_output = gzip_tallies, gzip_output
        _chk(_L.chn_batch_submit(self.h, C.byref(b)))
        self._fifo.append((n_reads, None))

    def wait_host(self):
        n, Cn = self._fifo[0][0], self.C
        out = dict(num_hashes=np.zeros(n, np.uint32), counts=np.zeros((n, Cn), np.uint32), unique=np.zeros((n, Cn), np.uint32),
                   probs=np.zeros((n, Cn), np.float64), call=np.zeros(n, np.uint8), conf=np.zeros(n, np.uint8),
                   flags=np.zeros(n, np.uint8))
        want_gz = len(self._fifo[0]) > 2 and self._fifo[0][2]
        gz_out = self._fifo[0][3] if len(self._fifo[0]) > 3 else 0
        if want_gz and gz_out != 1:
            out["gzip_tallies"] = np.zeros((n, 320), np.uint16)
        if want_gz and gz_out != 0:
            out["gzip_sizes"] = np.zeros(n, np.uint32)
        r = Result(C.sizeof(Result), 0, out["num_hashes"].ctypes.data, out["counts"].ctypes.data, out["unique"].ctypes.data,
                   out["probs"].ctypes.data, out["call"].ctypes.data, out["conf"].ctypes.data, out["flags"].ctypes.data,
                   out["gzip_tallies"].ctypes.data if "gzip_tallies" in out else None,
                   out["gzip_sizes"].ctypes.data if "gzip_sizes" in out else None)
        _chk(_L.chn_batch_wait(self.h, C.byref(r)))
        self._fifo.pop(0)
        return out

    def wait_device(self):
        r = Result()
        r.struct_size, r.on_device = C.sizeof(Result), 1
        _chk(_L.chn_batch_wait(self.h, C.byref(r)))
        self._fifo.pop(0)
        return r

    def sync(self):
        _chk(_L.chn_stream_sync(self.h))

    def classify_counts(self, num_hashes, counts, unique, lengths, mean_quality, compression):
        n, Cn = len(num_hashes), self.C
        a = [np.ascontiguousarray(x, dt) for x, dt in ((num_hashes, np.uint32), (counts, np.uint32), (unique, np.uint32),
                                                       (lengths, np.uint32), (mean_quality, np.float32), (compression, np.float32))]
        out = dict(probs=np.zeros((n, Cn), np.float64), call=np.zeros(n, np.uint8), conf=np.zeros(n, np.uint8))
        _chk(_L.chn_classify_counts(self.h, n, *[x.ctypes.data for x in a], out["probs"].ctypes.data, out["call"].ctypes.data,
                                    out["conf"].ctypes.data))
        return out

    def profile(self, which, reset=False):
        ms, n = C.c_double(), C.c_uint64()
        _chk(_L.chn_stream_profile(self.h, which, C.byref(ms), C.byref(n), 1 if reset else 0))
        return ms.value, n.value

    def last_batch_bytes(self):
        b, m = C.c_uint64(), C.c_uint64()
        _chk(_L.chn_stream_last_batch_bytes(self.h, C.byref(b), C.byref(m)))
        return b.value, m.value

    def destroy(self):
        if self.h:
            _L.chn_stream_destroy(self.h)
            self.h = None


def synth_genomes(device, seed, n_genomes, genome_len):
    p = C.c_void_p()
    _chk(_L.chn_synth_genomes(device, seed, n_genomes, genome_len, C.byref(p)))
    return p.value


def synth_reads(device, seed, dev_genomes, n_genomes, genome_len, n_reads, len_min, len_max, sub_rate=0.05,
                random_fraction=0.1, mean_quality=40.0, first_read_id=0):
    out = SynthReadsOut()
    _chk(_L.chn_synth_reads(device, seed, dev_genomes, n_genomes, genome_len, first_read_id, n_reads, len_min, len_max, sub_rate,
                            random_fraction, mean_quality, C.byref(out)))
    return out


def pinned_array(shape, dtype):
    """numpy array backed by page-locked host memory (chn_host_alloc); keep the returned array alive while in use.
    The memory is released with host_free(arr)."""
    dtype = np.dtype(dtype)
    n = int(np.prod(shape))
    p = C.c_void_p()
    _chk(_L.chn_host_alloc(max(n * dtype.itemsize, 16), C.byref(p)))
    buf = (C.c_char * (n * dtype.itemsize)).from_address(p.value)
    arr = np.frombuffer(buf, dtype=dtype, count=n).reshape(shape)
    _PINNED[arr.ctypes.data] = p.value
    return arr


_PINNED = {}


def host_free(arr):
    p = _PINNED.pop(arr.ctypes.data, None)
    if p:
        _chk(_L.chn_host_free(p))


def device_malloc(device, nbytes):
    p = C.c_void_p()
    _chk(_L.chn_device_malloc(device, nbytes, C.byref(p)))
    return p.value


def device_upload(device, ptr, arr):
    arr = np.ascontiguousarray(arr)
    _chk(_L.chn_device_upload(device, ptr, arr.ctypes.data, arr.nbytes))


def device_free(device, ptr):
    if ptr:
        _chk(_L.chn_device_free(device, ptr))


def device_download(device, ptr, nbytes, dtype):
    out = np.zeros(nbytes // np.dtype(dtype).itemsize, dtype)
    _chk(_L.chn_device_download(device, out.ctypes.data, ptr, nbytes))
    return out
