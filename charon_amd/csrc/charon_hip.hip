// libcharon_hip.so -- MI355X (gfx950 / CDNA4) implementation of the per-read classification path of
// `charon dehost` behind the C ABI of include/charon_hip.h.
//
// Kernel chain per batch (main stream: minimise+probe; side stream: counts, model+call; copy stream: uploads + ordering; a stream for the
// long reads' launch; up to three batches in flight, no host round trip):
//   k_len_hist/scatter        order reads by length class, longest first, so that a wavefront holds reads of similar length; publishes how many
//                             reads are "long" (32 768 bases and more)
//   k_minimise_probe          ONE LANE PER READ: rolls the canonical base-5 k-mer hash (seqan3 minimiser_hash semantics,
//                             src/dehost_main.cpp:317-318,367), runs the exact sequential window-minimum emission rule (ties and
//                             homopolymers are exact by construction), compacts the emitted minimisers of the 64 reads of the
//                             wavefront into an LDS queue, and every 64 queued minimisers does one full-width probe round:
//                             64 lanes x h gathers of W words from the HBM-resident interleaved Bloom filter (bulk_contains,
//                             src/dehost_main.cpp:368), AND, then either accumulates per-category hit / unique-hit counters in
//                             LDS (FUSED: every category owns one bin, C <= 8; indexes of at most four bins fetch one row at a time and
//                             stop when the AND is empty) or appends a compact entry to the wavefront's row log in HBM
//     ... its SPLIT launch    ONE WAVEFRONT PER LONG READ: 64 pieces overlapping by w - 1 bases, a piece starts cold only where the window
//                             minimum is unique (exact seams), all lanes add to one owner slot
//   k_count_wavelog           (general layouts) one workgroup per wavefront log: per-bin totals -> first max bin per category ->
//                             unique hits  (ReadEntry::get_counts, include/read_entry.hpp:92-138); sized to run beside the next batch's
//                             probe wavefronts; k_count_longlog does the same for the logs of the SPLIT launch
//   k_model_call              one lane per read: KDE/dexp probability (memoised, misses evaluated by the whole wavefront) +
//                             call_host / call_category (include/classify_stats.hpp:242-252,370-389; include/read_entry.hpp:157-279)
// The source is one translation unit; the parts under parts/ are included in order at the bottom of this file.
// No MFMA: the path is integer/bit work bound by random 8-16 byte gathers from HBM.
//
// This file is written for gfx950 only.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <functional>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include "charon_hip.h"
#include "default_kde.inc"

#include "parts/common.inc"
#include "parts/k1_minimise_probe.inc"
#include "parts/k2_count_wavelog.inc"
#include "parts/shard_kernels.inc"
#include "parts/shardx_kernels.inc"
#include "parts/k3_model_call.inc"
#include "parts/len_order.inc"
#include "parts/gzip_tally.inc"
#include "parts/gzip_size_dev.inc"
#include "parts/ef_decode.inc"
#include "parts/synth_kernels.inc"
#include "parts/abi_index_model.inc"
#include "parts/abi_stream_batch.inc"
#include "parts/abi_shard_wait.inc"
#include "parts/abi_shardx.inc"
#include "parts/abi_synth_memory.inc"
