/*
 * knn_mi355x.h — C-ABI of libknn_mi355x.so: brute-force nearest-neighbour
 * search (1-NN, squared L2, fp32, first-minimum tie-break) on MI355X (gfx950).
 *
 * This is the drop-in boundary for ONE hot path of wu-kan/multicore-hw2: the
 * global cudaCallback() of reference sources/src/core.h:71 (defined at
 * sources/src/core.cu:1282-1297, which forwards to v8::cudaCallback,
 * core.cu:856-958).  Plain pointers and sizes only; no HIP, torch or C++ types.
 *
 * Results are bit-identical to the reference's serial v0 path
 * (core.cu:27-62): squared distance accumulated in fp32 in dimension order,
 * one rounding per operation (no FMA), and the lowest index among equal
 * minima.
 *
 * There is NO CPU fallback in this library.  Where the reference silently
 * computes on the CPU when no GPU is present (core.cu:869-870), this library
 * reports the error and exits like the reference's CHECK macro does for any
 * other runtime error (core.h:77-87).
 */
#ifndef KNN_MI355X_H
#define KNN_MI355X_H

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------
 * 1. The drop-in entry point.  Replaces: core.h:71 / core.cu:1282-1297.
 *
 *   k, m, n            dimensions, #queries, #references (all >= 1)
 *   searchPoints       host, fp32, row-major [m][k]   (read-only, not freed)
 *   referencePoints    host, fp32, row-major [n][k]   (read-only, not freed)
 *   *results           set to a malloc()'d int[m]; caller free()s it
 *                      (core.cu:35,59,935; main.cu:98,175)
 *
 * results[j] = 0-based index of the reference nearest to query j.
 * Synchronous: all device work has finished on return.  The reference set is
 * split over every visible GPU in contiguous index ranges (the scheme of
 * core.cu:873-883) and the per-GPU winners are min-reduced as packed
 * (distance, global index) keys.  On a runtime error prints
 * "Error: <file>:<line>, code:<c>, reason: <text>" and exit(1)s (core.h:77-87).
 * ---------------------------------------------------------------------- */
void cudaCallback(int k, int m, int n, float *searchPoints, float *referencePoints,
                  int **results);

/* The host-input entry points (cudaCallback, knn_index_query_host, knn_index_create from host
 * rows) keep their device staging buffers in a small per-device pool between calls — hipMalloc +
 * hipFree cost more than the scan at the reference's test sizes.  At most 64 buffers / 4 GiB per
 * device; knn_trim() gives them all back to the runtime (returns the number of bytes released). */
long long knn_trim(void);

/* ------------------------------------------------------------------------
 * 2. Device-resident index API (not in the reference; cudaCallback is
 *    create + query + destroy over all GPUs).  Lets a caller keep the
 *    reference set in HBM across calls and time the kernels without PCIe.
 *    All functions return 0 on success, a negative KNN_E* code on failure;
 *    knn_last_error() gives the message for the calling thread.
 * ---------------------------------------------------------------------- */
typedef struct knn_index knn_index;

enum {
    KNN_OK = 0,
    KNN_EINVAL = -1,   /* bad argument */
    KNN_ENODEV = -2,   /* no usable GPU */
    KNN_EHIP = -3,     /* HIP runtime error */
    KNN_ENOMEM = -4
};

/* Packed result key: (float_bits(dist2) << 32) | global_index.  dist2 >= +0 so
 * unsigned order == lexicographic (distance, index) order == v0's first strict
 * minimum.  KNN_KEY_INIT = (+INF, index 0): what v0 returns when nothing
 * beats +INF (core.cu:39-40,50). */
#define KNN_KEY_INIT 0x7F80000000000000ull

int knn_device_count(void);
const char *knn_last_error(void);
const char *knn_version(void);

/* Build an index over one contiguous shard of the reference set on `device`.
 *   refs           fp32 row-major [n_local][k]; host pointer if
 *                  refs_on_device == 0 (copied to the GPU), else a device
 *                  pointer on `device` that must stay valid until destroy
 *                  (borrowed, not copied)
 *   base_index     global index of refs[0] (the shard offset that
 *                  core.cu:932-933 adds on the host; here it is folded into the
 *                  keys on the GPU)
 *   stream         hipStream_t as void* (NULL = default stream) used for the
 *                  layout-preparation kernels
 * n_local may be 0 (an empty shard: queries leave the keys untouched). */
int knn_index_create(knn_index **out, int device, int k, long long n_local, const float *refs,
                     int refs_on_device, long long base_index, void *stream);
void knn_index_destroy(knn_index *idx);

/* ------------------------------------------------------------------------
 * 2b. Cell-range shards (round 4): how a caller with several GPUs (one process per GPU, or one index per GPU in one process)
 *     splits the reference set so that the GPUs' scans ADD UP to the scan of one GPU holding everything.
 *     The reference splits by index range (core.cu:875-883) and so does cudaCallback here — rows arrive in the caller's
 *     order.  For a resident set that costs work: each range is gridded by itself at 1 / N of the resolution, and the ranks
 *     together do 3.8x the tile steps of one GPU at N = 8 (profiles/r04_shard_sim.txt).  Instead:
 *       1. every rank takes a strided sample of its rows; the samples are gathered (a few hundred KB) and
 *          knn_geom_create builds ONE grid from them — identical input, identical grid on every rank;
 *       2. knn_geom_assign says which rank's cell range each row falls into; the caller moves the rows there (one
 *          all-to-all at build time, not on the query path) together with their global numbers, ascending per rank;
 *       3. knn_index_create_sharded sorts a rank's rows into ITS cells of the global grid;
 *       4. knn_index_seed_export writes the first few tiles of each of the rank's cells into its part of a buffer of
 *          knn_geom_info()[5] bytes; the caller all-gathers the parts and hands the whole layer to knn_index_seed_attach
 *          (replicated: 151 MB for 2^16 cells).  A query bounds its answer from the 4 seed cells around it (16 measured slower, profiles/r04_seed_sweep.txt) — this rank's
 *          whole, the others' through the layer — so a rank prunes almost as if it saw everybody's rows.
 *     Queries, keys and the exchange step (min over the ranks' keys) are the same calls as for index-range shards: the
 *     keys carry global row numbers (gids) when the batch's last kernel has run.
 * ---------------------------------------------------------------------- */
typedef struct knn_geom knn_geom;
/* sample_host: samples x k floats, rows of the GLOBAL set (>= 64; a few thousand is plenty); seed_tiles: tiles of every cell
 * in the replicated layer (0 = 2).  KNN_EINVAL when the set does not suit (k > 16, too few rows per rank, degenerate sample). */
int knn_geom_create(knn_geom **out, int k, long long n_global, int nranks, const float *sample_host, long long samples,
                    int seed_tiles);
void knn_geom_destroy(knn_geom *g);
/* out = {bits of the global grid, its cells, cells per rank, seed tiles per cell, bytes of one rank's part of the seed layer,
 * bytes of the whole layer, bits of the low pruning table, ranks} */
int knn_geom_info(const knn_geom *g, long long out[8]);
/* First cell code of `rank`'s range (rank = the number of ranks: the number of cells); ranges start at multiples of 2^sa. */
long long knn_geom_first_cell(const knn_geom *g, int rank);
/* owner_dev[i] = rank whose cell range holds rows_dev[i] (device arrays on `device`; synchronous on return). */
int knn_geom_assign(const knn_geom *g, int device, const float *rows_dev, long long n, int *owner_dev, void *stream);
/* refs_dev: the rank's rows (device, borrowed); gids_dev[i]: global row number of refs_dev[i], STRICTLY ASCENDING, < 2^31
 * (device, borrowed) — v0's lowest-index tie-break is decided by local row order.
 * LIFETIME: refs_dev, gids_dev and the layer given to knn_index_seed_attach are BORROWED for the index's whole life: the prep,
 * scan and finalise kernels of every later query read them.  They must stay allocated and unchanged until knn_index_destroy
 * returns (a caller that lets them go gets use-after-free reads inside those kernels).  KNN_EINVAL when a row lies outside the
 * rank's cell range of the geometry.  The index must be given the seed layer (export on every rank, gather, attach) before
 * it is queried by more than one rank's worth of queries: without it the bound comes from the rank's own cells only —
 * still exact, only slower. */
int knn_index_create_sharded(knn_index **out, int device, const knn_geom *g, int rank, long long n_local, const float *refs_dev,
                             const unsigned *gids_dev, void *stream);
int knn_index_seed_export(knn_index *idx, void *layer_dev, void *stream);   /* this rank's part, in place in the whole-layer buffer */
int knn_index_seed_attach(knn_index *idx, const void *layer_dev);           /* the gathered layer (borrowed until destroy) */

/* Fill keys_dev[0..m) with KNN_KEY_INIT (async on stream). */
int knn_keys_init(int device, unsigned long long *keys_dev, int m, void *stream);

/* keys_dev[j] = min(keys_dev[j], key of the nearest reference of this shard to
 * query j).  queries_dev: device fp32 [m][k].  Asynchronous on `stream`.
 * Several shards (or GPUs, after an all-reduce MIN) may fold into one key
 * array; the minimum is the global answer. */
int knn_index_query_keys(knn_index *idx, int m, const float *queries_dev,
                         unsigned long long *keys_dev, void *stream);

/* Same, using query workspace `slot` (0 .. 7) of the index.  The index owns eight independent
 * workspaces, so up to eight batches may be in flight at once on their own streams (e.g. batch
 * i+1's small preparation kernels beside batch i's scan); calls that share a slot must be
 * stream-ordered by the caller.  Calls on one index from several host threads are serialised by the
 * index (a lock around the ~20 us of host work of a call; the GPU work of different slots still
 * overlaps); knn_index_last_stats then describes whichever call was enqueued last.  A slot outside
 * 0 .. 7 is KNN_EINVAL.
 * knn_index_query_keys == slot 0. */
int knn_index_query_keys_slot(knn_index *idx, int slot, int m, const float *queries_dev,
                              unsigned long long *keys_dev, void *stream);

/* Same, with flags.  KNN_QUERY_INIT_KEYS: the call WRITES the keys — (+INF, 0) min-folded with this shard's
 * answer — instead of folding into what keys_dev holds: the caller needs no knn_keys_init launch in front (the
 * cell-pruned path sets the keys inside its first kernel; the other paths issue the fill themselves). */
#define KNN_QUERY_INIT_KEYS 1u
int knn_index_query_keys_ex(knn_index *idx, int slot, int m, const float *queries_dev,
                            unsigned long long *keys_dev, void *stream, unsigned flags);

/* Same, and ALSO the int32 indices of the batch: indices_dev[j] = (int)(keys_dev[j] & 0xFFFFFFFF) once the shard's answer has
 * been folded in (indices_dev may be NULL: then exactly knn_index_query_keys_ex).  For a caller whose answer is this one
 * shard's (one GPU holds the whole set): on the cell-pruned path the indices are written by the last block of the batch's
 * last kernel, so the batch costs no knn_keys_to_indices launch; the other paths issue that launch themselves.  Callers that
 * merge several shards' keys first (min over shards / GPUs) pass NULL and unpack after the merge. */
int knn_index_query(knn_index *idx, int slot, int m, const float *queries_dev, unsigned long long *keys_dev,
                    int *indices_dev, void *stream, unsigned flags);

/* The path's one exchange step, for callers that keep one index per GPU in ONE process: min-reduce the
 * GPUs' key arrays with RCCL — ncclAllReduce(ncclUint64, ncclMin) per device inside
 * ncclGroupStart/End over xGMI, communicators from ncclCommInitAll created ONCE per process: the first call
 * fixes the device list, a later call with another list returns KNN_EHIP (merge those keys on the host).
 * Replaces the reference's host gather + CPU re-rank (core.cu:925-957).
 *   devices[g]    HIP device of key array g (each device at most once)
 *   keys_dev[g]   m packed keys on devices[g]; on completion every array holds the elementwise
 *                 unsigned minimum (= lexicographic (distance, index) minimum) of all of them
 *   streams[g]    hipStream_t as void* the reduction of array g is enqueued on (NULL array or NULL
 *                 entries = default stream); stream-ordered after the queries that wrote the keys
 * One process per GPU (bench.py) uses torch.distributed's all_reduce(MIN) on the same keys instead.
 * librccl is opened at first use; KNN_EHIP with the reason if it cannot be. */
int knn_keys_allreduce_min(int ndev, const int *devices, unsigned long long *const *keys_dev, int m,
                           void *const *streams);

/* out_dev[j] = (int)(keys_dev[j] & 0xFFFFFFFF) (async on stream). */
int knn_keys_to_indices(int device, const unsigned long long *keys_dev, int m, int *out_dev,
                        void *stream);

/* Convenience: host queries in, host indices out, synchronous. */
int knn_index_query_host(knn_index *idx, int m, const float *queries_host, int *out_host);

/* Tuning / test hooks.  Known names:
 *   "path"    0 = auto, 1 = exact VALU kernels only, 2 = force the MFMA filter
 *             (+ exact re-rank) where its preconditions hold, 3 = the uniform-grid spatial index
 *             where k <= 4 and the data allows it (else the exact kernels).  Auto: resident shards
 *             with k <= 4 and >= 16384 rows get the grid index at creation and are served by it
 *   "shards"  cudaCallback only: split the reference set into this many
 *             shards.  0 = the library's rule: ONE GPU when n <= min(2^18, m << 10) (the reference's rule,
 *             core.cu:871-872: every case of the TA harness), else as many of the visible GPUs as the cost
 *             model says repay their fan-out (knn_debug_shard_policy; knn_get_option("last_shards") tells what the
 *             most recent call used).  Shards beyond the GPU count wrap around the devices — exercises the
 *             partition + merge logic on a single GPU.
 *   "filter_qt" tuning: query tiles (of 32) each filter wave keeps in registers: 8, 16 or 32
 *             (0 = chosen from m)
 *   "stream"  cudaCallback only: scan each shard chunk by chunk under its host-to-device copy
 *             with the exact kernels: 0 = when the cost model says so, 1 = never, 2 = always
 *             (shards of at least 64 MiB)
 *   "ingest"  indexes created from HOST rows (knn_index_create with refs_on_device = 0, and the
 *             staged-filter case of cudaCallback): 0 = the rows go over in chunks and every chunk's
 *             MFMA layouts are built as soon as it has landed (robust box from a strided host
 *             sample), 1 = copy everything, then build (the box from full-range statistics)
 *   "rccl"    cudaCallback only: how the shards' keys are merged: 0 = RCCL all-reduce when the set
 *             is split over ALL visible GPUs (one shard each), host merge otherwise (a process holds ONE
 *             communicator set, for all its GPUs: knn_get_option("rccl_comm_sets") is 0 or 1); 1 = RCCL
 *             whenever the shards are one per visible GPU (also with one GPU: a 1-rank communicator);
 *             2 = host merge always.
 *             knn_get_option("rccl_reductions") counts the merges RCCL has done,
 *             knn_get_option("rccl_version") is the loaded library's NCCL_VERSION_CODE (0: none)
 *   "cells"   the MFMA filter's cell-pruned form (k <= 32): the index sorts the shard into 2^B cells (every
 *             dimension — the first 16 when k > 16 — cut at sample quantiles), and a batch scores only the cells each query
 *             could not rule out by its distance to the cell's box.  0 = library policy: indexes created with
 *             knn_index_create of >= 2^19 rows (k <= 12), >= 2^20 rows (k = 13 .. 16), >= 2^22 rows (k = 17 .. 21), >= 2^23 rows
 *             (k = 22, 23) or >= 2^24 rows (k = 24, 25; beyond that too few cells are ruled out for the pruned scan to win); for the one-shot
 *             cudaCallback on shards of that size when the cost model says the batch repays the sort (the bucket pass of
 *             the sort runs under the host-to-device copy, ~1 ms per 2^24 rows stays behind the last byte: from about
 *             2500 queries on at n = 2^24; knn_get_option("last_cells") counts the shards of the most recent call
 *             that the pruned scan served).  1 = every index of >= 2^17 rows (k <= 32),
 *             cudaCallback's shards included; 2 = never.  Read when an index is created; 2 also makes
 *             existing indexes use the full scan.  Any distribution keeps its cells: a cell of many rows
 *             (clustered, low-rank data) is cut into several work items, empty cells cost nothing.  A batch
 *             the pruned path cannot bound (non-finite or far-away queries, no reference row found to bound a
 *             query with) is answered by the exact scan of the shard; a batch whose candidates overflow the
 *             record buffers (rows of a cluster tighter than the fp16 step) by the exact arithmetic over its
 *             listed (cell, query) pairs only; the next batch is back on the pruned path.  Results are
 *             bit-exact either way
 *   "scan_blocks" the pruned scan's blocks per CU: 0 = auto (two; one for shards of up to 2^15 cells while the index's last
 *             eight calls named more than one workspace slot — batches in flight side by side: the scan alone gets 10-20 %
 *             longer and the next batch's preparation kernels find room beside it, 5-7 % per step), 1, 2
 *   "run_thresholds" the deep-K scans (64 < k <= 512): 0 / 1 = every score below a query's threshold lowers that threshold for
 *             the rest of the launch (threshold' = max(floor, score + margin), shared between blocks through an atomic minimum
 *             per query; the sample pass then only visits every 32nd tile), 2 = thresholds stay what the sample pass made them
 *   "sample_stride" deep-K scans: tiles the sample pass skips between two it scores (0 = library policy)
 *   "cells_centre" per-cell frames of the cell-sorted layout (k <= 16, not for cell-range shards): the fp16 fragments of a cell are
 *             taken about the middle of the cell's own box and scaled (by up to 2^8 more) to fill the fp16 range the rows of the
 *             whole shard share otherwise — the filter's rounding error shrinks from 2^-12 of the shard's box to 2^-12 of the
 *             cell's, what a set of tight clusters needs for its thresholds to separate anything.  0 = library policy: when the
 *             build's sample of the rows looks clustered (median nearest-neighbour distance inside the sample below 1/16 of the
 *             box); 1 = every cell-sorted layout; 2 = never.  Read when an index is built.  Costs one more pass over the rows at
 *             build time and ~100 instructions per (cell, 32 queries) at query time; results are identical either way.
 *             knn_get_option("cells_centred_builds") counts the layouts built so (read-only).
 *   "cells_lists" who makes a cell's list of queries (those of the batch that cannot rule the cell out) on the pruned path:
 *             1 = knn_cells_match_kernel in a launch of its own between the preparation and the scan (lists in memory),
 *             2 = the scan's waves for the items they take (same test, same arithmetic, lists in LDS: one launch and one
 *             dependent round trip per item fewer), 0 = auto: 2 for shards of up to 2^13 cells queried one batch at a time
 *   "scan_deal" how the pruned scan's waves get their work items (runs of tiles of one cell): 1 = fixed (wave w takes items
 *             w, w + W, ...), 2 = a block owns a contiguous run and its waves take items from a counter in LDS (the launch
 *             is 6-8 % shorter: no wave is left with twice the average), 0 = auto: 2 for callers that query one batch at a
 *             time on shards with at least two items per wave, and for shards of more than 2^15 cells always; 1 below two items
 *             per wave, and on the smaller shards while the last eight calls on the index named more than one workspace slot
 *             (batches in flight fill each other's gaps; the fixed deal's cheaper prologue then gives the shorter step).  The automatic choices follow the caller's recent behaviour: a
 *             caller that goes back to one batch at a time gets the one-batch shapes again after eight calls
 *   "cells_build" how the cell-sorted layout is built: 0 = two passes (rows grouped into 256 buckets of consecutive cells, then
 *             placed bucket by bucket out of one XCD's L2; needs n x 72 bytes of scratch + 1/8: used for shards of up to 2^25 rows,
 *             and falls back when the scratch does not fit) in their FAST form — buckets of fixed room, no counting pass over the
 *             rows, tile ranges and work items from a device prefix, one synchronisation for the whole build; a bucket that
 *             outgrows its room (rows the cuts do not spread) makes the build start over in the counted form; 1 = the one-pass
 *             placement; 2 = the counted two-pass build (rounds 3-4: a pass for the bucket counts, two host round trips).
 *             Read when an index is created
 *   "filter_rounds" tuning: filter workgroups per resident slot (0/1 = one: persistent waves)
 *   "filter_chain" filter scans issued on different workspace slots / streams: 1 = run one
 *             after the other (event-chained), 2 = free to overlap, 0 = auto (chained when the
 *             shard is >= 16M references: a launch's duration then stays that of the kernel alone)
 * Returns KNN_EINVAL for an unknown name or value. */
int knn_set_option(const char *name, long long value);
long long knn_get_option(const char *name);

/* Statistics of the most recent knn_index_query_keys on this index (filled
 * when the stream has completed; call after synchronising):
 *   [0] path taken (1 exact, 2 filter, 3 grid index, 4 filter in its cell-pruned form)
 *   [1] candidates re-ranked exactly
 *   [2] how the device answered a batch the filter could not: 0 = it could; 1 = exact scan of the whole shard (a query
 *       nothing bounds: not finite, far outside the references' box); 2 = cell-pruned path only: more candidates than the
 *       record buffers hold (rows of a cluster tighter than the fp16 step), the batch's listed (cell, query) pairs were
 *       evaluated with the exact arithmetic — the cells the geometry ruled out stay ruled out
 *   [3] reference rows outside the filter's robust box (scanned exactly on every query) */
int knn_index_last_stats(knn_index *idx, long long stats[4]);

/* Test / development hook: counters of the most recent batch on the cell-pruned path (call after synchronising; all 0
 * on other paths): [0] queries whose seed cells held no reference row (their bound came from a strided sample of the
 * layout), [1] cells whose query list outgrew its on-chip room (scored against the whole batch instead),
 * [2] cells of the index, [3] rows of its largest cell. */
int knn_index_debug_counters(knn_index *idx, long long out[4]);

/* Test hook (host arithmetic only, no GPU needed): the number of GPUs a cudaCallback(k, m, n, ...) is split over on a
 * node with ndev visible devices — the reference's rule (core.cu:865-872) plus the library's cost model. */
int knn_debug_shard_policy(int k, int m, long long n, int ndev);
/* Test hook (host arithmetic only): how ONE shard of `rows` rows of a one-shot cudaCallback(k, m, ...) would be served under the
 * current options: out = {filter layouts (0 none: exact kernels, 1 plain, 2 cell-sorted: the pruned scan), 1 if the exact scan
 * runs chunk by chunk under the copy, 1 if the grid index (k <= 4) serves it, copy calls of the streamed form}. */
int knn_debug_plan_shard(int k, int m, long long rows, long long out[4]);

/* Test hook (host arithmetic only, no GPU needed): every size one scan launch of the cell-pruned path and the re-rank behind
 * it index with, for an index of `nitems` work items on a device of num_cu CUs and a batch of m <= 1024 queries:
 * out = {scan blocks, record lists (= scan waves), records per list, first record of the shared overflow area, its
 * capacity, dynamic LDS bytes of the scan, records a workspace holds, list counters a workspace holds}. */
int knn_debug_scan_plan(int num_cu, int blocks_per_cu, unsigned nitems, int m, long long out[8]);
/* The same with the list maker named: self_lists != 0 = the scan's waves list their own items (option "cells_lists" 2): the
 * dynamic LDS then also holds the batch's Dup values and one list room per wave. */
int knn_debug_scan_plan_ex(int num_cu, int blocks_per_cu, unsigned nitems, int m, int self_lists, long long out[8]);

/* Test hook for the filter's error bound: raw MFMA filter scores S[m][n_local] (row-major,
 * device) for a query batch, the fp32 squared norms M[m] of the fp16 query rows (device), and
 * consts = {sigma, eta, rho, Amax, Bmax, g2, gamma, #non-finite fp16 query coordinates}.  A pair's
 * score obeys |S + M - sigma^2 d^2| <= 2 eta sigma d + eta^2 + rho (+ gamma M), d = real distance.
 * Synchronous; needs an index that has filter layouts in row order (else KNN_EINVAL / KNN_EHIP: a cell-sorted
 * index — option "cells" — has no [m][n_local] score matrix). */
int knn_debug_filter_scores(knn_index *idx, int m, const float *queries_dev, float *scores_dev,
                            float *qnorm_dev, double consts[8]);

/* Bench support: time the index's dominant kernel (the one the roofline is quoted for) with a
 * HIP event pair recorded on the caller's stream around the launch.  enable = N > 0 brackets every
 * N-th later knn_index_query_keys (1 = all; an event record costs the stream ~5 us, which shows
 * in sub-100-us steps), 0 stops and drops the record. */
int knn_index_timing(knn_index *idx, int enable);
/* Waits for the recorded events, returns how many launches were timed and their summed
 * duration in milliseconds, and clears the record. */
int knn_index_timing_read(knn_index *idx, int *launches, double *total_ms);

/* Bench support: x[i] = (float)((splitmix64-style hash of (seed, first + i)) >> 40) * 2^-24,
 * uniform in [0,1), written on the device (same values as the oracle's
 * knn_synth_fill for the same seed/first). */
int knn_synth_fill_device(int device, float *dst_dev, long long count, unsigned long long seed,
                          long long first, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* KNN_MI355X_H */
