// knn_common.h — shared declarations of the knn_mi355x library internals (gfx950 only).
#pragma once
#include <atomic>

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>

#define KNN_WAVE 64

typedef unsigned long long u64;

static constexpr u64 kKeyInit = 0x7F80000000000000ull;  // (+INF, index 0)

static inline int knn_divup(long long a, long long b) { return (int)((a + b - 1) / b); }

// ---- exact VALU kernels (knn_exact.hip) -----------------------------------
// Folds the nearest reference of refs[0..n_local) (global index base + i) for each of the m
// queries into keys[] with a 64-bit unsigned atomic min.  Every distance is computed with the
// v0 arithmetic (reference core.cu:44-49).  If `gate` is non-null the kernels return at once
// unless *gate != 0 (device-side fallback switch, so the query path never syncs with the host).
hipError_t knn_exact_launch(int k, int m, long long n_local, long long base, const float *q_dev,
                            const float *r_dev, u64 *keys_dev, int num_cu, const unsigned *gate,
                            hipStream_t stream);

// Exact scan of the rows listed in list_dev[0..count) (global index = base + row).
hipError_t knn_exact_gather_launch(int k, int m, unsigned count, long long base, const float *q_dev,
                                   const float *r_dev, const unsigned *list_dev, u64 *keys_dev, int num_cu,
                                   const unsigned *gate, hipStream_t stream);

// Exact re-rank of the filter's candidate records (see knn_rerank_kernel).
// Record lists [list_base[i], list_base[i+1]) belong to piece i of the batch, whose records number their
// queries from qrow_base[i] (see plan_pieces in knn_filter.hip); unused entries have list_base = ~0.
struct RerankPieces {
    unsigned list_base[4] = {0u, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    unsigned qrow_base[4] = {0u, 0u, 0u, 0u};
    int n = 1;
};

// perm (nullable): records name positions of a permuted layout; perm[position] = row (~0u = padding), and n
// is then the number of positions.
// ovf_cap != 0: records[ovf_base ..) hold ctl[KNN_CTL_RECORDS] more records (capped at ovf_cap) that belong to no list.
hipError_t knn_rerank_launch(int k, long long n, const float *q_dev, const float *r_dev, long long base,
                             const u64 *records, const unsigned short *record_rows, const unsigned *counts,
                             unsigned nlists, unsigned slice, unsigned *ctl, u64 *keys, RerankPieces pieces,
                             hipStream_t stream, const unsigned *perm = nullptr, unsigned ovf_base = 0u,
                             unsigned ovf_cap = 0u);

hipError_t knn_keys_fill_launch(u64 *keys_dev, int m, hipStream_t stream);
hipError_t knn_keys_unpack_launch(const u64 *keys_dev, int m, int *out_dev, hipStream_t stream);
hipError_t knn_synth_fill_launch(float *dst, long long count, u64 seed, long long first,
                                 hipStream_t stream);

// ---- MFMA filter + exact re-rank (knn_filter.hip) ---------------------------
// Device-side control words of one filter query (FilterState::ctl).
enum {
    KNN_CTL_FALLBACK = 0,  // != 0: the exact kernels must scan the whole shard (filter unusable)
    KNN_CTL_RECORDS = 1,   // records appended to the SHARED overflow area behind the waves' slices (cell-pruned path)
    // words 2..4 are the out[0..2] window of knn_frag_kernel for the query batch
    KNN_CTL_AMAX = 2,      // float bits: max |scaled query coordinate| in fp16
    KNN_CTL_QNMAX = 3,     // float bits: max fp32 squared norm of the fp16 query rows
    KNN_CTL_QBAD = 4,      // != 0: a query coordinate is non-finite or out of fp16 range
    KNN_CTL_CELLS = 5,     // != 0: this batch went through the cell-pruned scan
    KNN_CTL_WIDE_SEEDS = 6,  // cell-pruned path: queries whose seed cells held no row (bounded by a strided sample instead)
    KNN_CTL_DENSE_CELLS = 7, // cell-pruned path: cells whose query list outgrew its LDS room (scored against the whole batch)
    KNN_CTL_EXACT_CELLS = 8, // cell-pruned path, != 0: more candidates than the record buffers hold (the fp16 scores cannot tell
                             // the rows of a tight cluster apart): the listed (cell, query) pairs are evaluated exactly instead
    KNN_CTL_SCAN_DONE = 9,   // cell-pruned path: blocks of the scan that have finished (the last one finalises a clean batch)
    KNN_CTL_TAIL_DONE = 10,  // cell-pruned path: blocks of the tail kernel that have finished
    KNN_CTL_DEFERRED = 11,   // cell-pruned path, != 0: some wave of the scan left a long record list to the tail kernel
    KNN_CTL_TOTAL = 12,      // cell-pruned path: records the scan's waves have published so far (steps of 64 per wave)
    KNN_CTL_WORDS = 13
};

#define KNN_SLOTS 8  // independent query workspaces per index: up to eight batches may be in flight
#define KNN_RECORD_CAPACITY (1u << 22)   // records a workspace holds (FilterWorkspace::records: 8 B each + 2 B row mask)
#define KNN_MAX_LISTS (1u << 16)         // record lists (= scan waves) a workspace has counters for (FilterWorkspace::counts)

// Per-batch scratch of the filter path (one per slot).
struct FilterWorkspace {
    int m_cap = 0;
    void *qry_frags = nullptr; // device [qtiles][kt][64] x 16 B: B operands (-2 * scaled query)
    float *qry_norms = nullptr;// device [qtiles*32]
    float *qry_amax = nullptr; // device [qtiles*32]: max |scaled fp16 coordinate| of each query
    float *thr = nullptr;      // device [4][qtiles*32]: thresholds | margins | floors | running thresholds (ordered uints) — the
                               // last three feed the deep-K scan's in-launch tightening (knn_filter.hip, "running thresholds")
    unsigned *ctl = nullptr;   // device [3][KNN_CTL_WORDS]: block 0 = the full-scan path (reset by its fragment kernel);
                               // blocks 1, 2 alternate between the batches of the cell-pruned path, whose first
                               // kernel clears the block the NEXT batch will use (no reset launch, no race with
                               // the flags its own waves raise)
    unsigned *ctl_cur = nullptr; // the block the most recent batch on this slot used (statistics)
    unsigned cell_batches = 0; // cell-pruned batches issued on this slot (picks the ctl block)
    u64 *records = nullptr;    // device [rec_cap]: nlists slices of `slice` records, one per wave
    RerankPieces pieces;
    bool has_rows = false;     // the last scan wrote a row mask next to every record
    unsigned rec_cap = 0;
    unsigned *counts = nullptr;// device [nlists]: records each wave produced (may exceed slice)
    unsigned nlists = 0, slice = 0;
    unsigned ovf_base = 0, ovf_cap = 0; // cell-pruned path: records [ovf_base, ovf_base + ovf_cap) take what a wave's slice
                               // cannot hold (a thousand copies of one query all hit the same tile); 0 = none
    float *umin = nullptr;     // device [sample blocks][m_padded]: per-block minima of the sample pass
    size_t umin_cap = 0;       // floats allocated in umin
    unsigned *qpart = nullptr; // device [3 * query blocks]: {max |coord|, max norm, #bad} per block
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;  // optional: bracket the filter kernel
    // cell-pruned scan (CellIndex): per-cell query lists and the per-query pruning tables of the batch
    unsigned *cell_counts = nullptr;       // device [ncells]
    unsigned short *cell_lists = nullptr;  // device [ncells][cap]
    float *dup = nullptr;                  // device [m_cap]: largest scaled squared distance a candidate can have
    float *lo_tab = nullptr, *hi_tab = nullptr;  // device [m_cap][2^sa], [2^(bits-sa)][m_cap]
    int cell_m_cap = 0;
    bool last_used_cells = false;
};

// 16-wide K steps of the fp16 layouts for dimension k: 1, 2, 4, 8 (register / LDS-tiled scans), 16 and 32 (LDS-tiled scan with
// two / one block of queries per wave: the B operands of k = 512 fill a wave's registers), beyond that a multiple of 8 (K walked
// in chunks of 128 dimensions, knn_filter_chunked_kernel); 0 = no filter for this k.
#define KNN_FILTER_MAX_K 4096
#define KNN_CELL_ITEM_TILES 18     // two passes of the scan (CELL_TILES_PER_PASS = 9)
static inline int knn_kt_of(int k)
{
    return k < 1 ? 0 : k <= 16 ? 1 : k <= 32 ? 2 : k <= 64 ? 4 : k <= 128 ? 8 : k <= 256 ? 16 : k <= 512 ? 32
         : k <= KNN_FILTER_MAX_K ? 8 * ((k + 127) / 128) : 0;
}

// Global geometry of a CELL-RANGE sharded set (round 4; include/knn_mi355x.h, knn_geom_*): ONE grid for the whole reference
// set — cuts, centre and scale from a sample of the global set, identical on every rank — whose cell codes are split into
// contiguous ranges, one per rank.  A rank's index sorts ITS rows into ITS cells of that grid, so the ranks' scans add up to
// the scan of one GPU holding everything (index-range shards re-grid n / N rows at 1 / N of the resolution and do 3.5x the
// work at N = 8: profiles/r04_shard_sim.txt).
#define KNN_SEED_HEADER_BYTES 256u
// Rank r of nranks owns the codes [first(r), first(r + 1)) of a grid of ncells cells, first(r) = r ncells / nranks rounded
// down to a multiple of gran (whole entries of the high pruning table; first(nranks) = ncells).
__host__ __device__ static inline unsigned knn_shard_first_cell(unsigned r, unsigned ncells, unsigned nranks, unsigned gran)
{
    if (r >= nranks)
        return ncells;
    return (unsigned)((unsigned long long)r * ncells / nranks) / gran * gran;
}
__host__ __device__ static inline unsigned knn_shard_owner(unsigned code, unsigned ncells, unsigned nranks, unsigned gran)
{
    unsigned r = (unsigned)((unsigned long long)code * nranks / ncells);
    if (r >= nranks)
        r = nranks - 1u;
    while (r > 0u && code < knn_shard_first_cell(r, ncells, nranks, gran))
        --r;
    while (r + 1u < nranks && code >= knn_shard_first_cell(r + 1u, ncells, nranks, gran))
        ++r;
    return r;
}
struct ShardGeom {
    int k = 0, bits = 0, sa = 0, nranks = 1;
    int seed_tiles = 2;              // tiles of every cell each rank replicates (the seed layer)
    unsigned char nb[16] = {0}, shift[16] = {0};
    unsigned ncells = 0;             // 2^bits, all ranks together
    unsigned cells_per_rank = 0;     // the LARGEST rank's cells (a part of the seed layer has room for that many)
    long long n_global = 0;
    float bounds[16 * 15];           // ascending cuts of every dimension (+INF beyond a dimension's bins)
    float center[16];
    float sigma = 1.0f;
    unsigned first_cell(int rank) const { return knn_shard_first_cell((unsigned)rank, ncells, (unsigned)nranks, 1u << sa); }
    unsigned cells_of(int rank) const { return first_cell(rank + 1) - first_cell(rank); }
    // bytes of ONE rank's part of the seed layer: header (bmax, nmax) | [cpr][T][64] fragments | [cpr][T][32] split norms
    size_t part_bytes() const { return KNN_SEED_HEADER_BYTES + (size_t)cells_per_rank * (size_t)seed_tiles * (1024u + 128u); }
};

// Cell-sorted layout of the references (k <= 16): see the head of knn_cells.hip.
struct CellIndex {
    int bits = 0, sa = 0;            // cells = 2^bits; low pruning table = 2^sa entries
    int lbits = 0;                   // bits of a local cell number (= bits unless the index is a cell-range shard)
    unsigned char nb[16] = {0}, shift[16] = {0};
    unsigned ncells = 0, cap = 0;    // cap: queries a cell's list can hold per batch
    float *bounds = nullptr;         // device [16][15]: ascending cuts of every dimension
    unsigned *tile_start = nullptr;  // device [ncells + 1]: first 32-row tile of each cell in the layout
    unsigned *perm = nullptr;        // device [ntiles * 32]: row held by each layout position (~0u = padding)
    unsigned max_cell_rows = 0;
    // what the scan's waves take one at a time: (cell << 48) | (tiles << 40) | first tile — a run of at most
    // KNN_CELL_ITEM_TILES tiles of ONE cell.  Cells without rows have no item; a cell of many rows (clustered data the
    // quantile cuts do not spread) has several, all scored against the same list
    unsigned long long *items = nullptr;
    unsigned nitems = 0;
    // build only (freed once the rows are placed): the shard's rows grouped by the top 8 bits of their cell code — 256
    // buckets of consecutive cells — as 64-byte records + (code << 32 | row); null: the one-pass placement is used
    float *tmp_rows = nullptr;
    unsigned long long *tmp_meta = nullptr;
    unsigned *bucket_start = nullptr;   // device [257]: first record of each bucket
    // the FAST build (round 5; knn_cells_build with fast = true): buckets of fixed room, filled to bucket_fill[b]; the cells' tile
    // ranges and the items come from a device prefix — build_res = {tiles, items, rows of the largest cell, bucket overflow}
    // is read by the caller together with the placement's statistics, ONE synchronisation for the whole build
    float h_bounds[16 * 15];            // (host sources of the fast build's asynchronous copies: they must outlive the call)
    unsigned h_bucket_start[257];
    unsigned bucket_cap = 0;            // records a bucket has room for
    unsigned *bucket_fill = nullptr;    // device [256] (inside the block `bucket_start` points to)
    unsigned *build_res = nullptr;      // device [4]   (likewise)
    // cell-range shards (knn_index_create_sharded): global number of every local row, ascending (borrowed); null: base + row
    const unsigned *gids = nullptr;
    // cell-range shards: `bits`, `nb`, `shift`, `sa`, `bounds` are the GLOBAL grid's; this index holds cells
    // [cell_base, cell_base + ncells) of it under local numbers 0 .. ncells - 1 (cell_base is a multiple of 2^sa)
    unsigned cell_base = 0;
    const ShardGeom *geom = nullptr;      // (borrowed: outlives the index)
    const unsigned char *seed_layer = nullptr;   // device: every rank's part, nranks x geom->part_bytes() (borrowed)
    // per-cell frames (round 5, clustered data; knn_cells_recentre): the fragments of cell c are fp16((row - centre_c) x scale_c)
    // instead of the shard's one frame — cell_frame[c] = { centre[16] (the rows' own units), scale_c = sigma 2^e, 2^e,
    // max |fragment coordinate| of the cell, max fragment norm of the cell }; tile_cell[t] = the cell tile t belongs to
    bool centred = false;
    float *cell_frame = nullptr;          // device [ncells][KNN_CELL_FRAME_WORDS]
    unsigned *tile_cell = nullptr;        // device [ntiles]
};
#define KNN_CELL_FRAME_WORDS 20
#define KNN_NIF_MAX_K 30   // 16 < k <= 30: the cell-sorted fragments carry the rows' norms in K-slots 30, 31 (knn_cells.hip: cell_tile_step_nif)

struct FilterState {
    bool usable = false;       // references finite and in a sane range: filter layouts exist
    int k = 0, kt = 0;         // real dimension; 16-wide K steps (padded k = 16 * kt)
    long long n = 0;           // references in the shard
    long long ntiles = 0;      // ceil(n / 32)
    float sigma = 1.0f;        // power-of-two scale
    float bmax = 0.0f;         // max |scaled fp16 reference coordinate|
    float nmax = 0.0f;         // max fp32 squared norm of the fp16 reference rows
    float *center = nullptr;   // device [16*kt]
    void *ref_frags = nullptr; // device [ntiles][kt][64] x 16 B: A operands in MFMA lane order
    float *ref_norms = nullptr;// device [ntiles*32] (+INF for padding rows)
    unsigned *ref_norms2 = nullptr; // cell-sorted layouts only, device [ntiles*32]: the same norms as two fp16 halves
                               // (hi | mid * 2^11 << 16) — the prep kernel rebuilds the C tile of its seed scores from
                               // them with one extra MFMA: one register per tile in flight instead of 16 (knn_cells.hip)
    int force_qt = 0;          // tuning hook: query tiles per wave (0 = pick by m)
    int force_rounds = 0;      // tuning hook: filter blocks per resident slot (0 = default)
    int chain_policy = 0;      // scans of different slots: 0 auto (chained when long), 1 always chained, 2 never
    unsigned *outliers = nullptr; // device: rows outside the robust box (excluded from the filter, scanned exactly)
    unsigned n_outliers = 0;
    CellIndex *cells = nullptr;   // non-null: the layout is cell-sorted (ntiles counts its padded tiles)
    int cells_policy = 0;         // per call: 0 use the cells when present, 2 full scan
    bool several_slots = false;   // a query has used a workspace slot other than 0: batches are in flight side by side
    int scan_deal = 0;            // pruned scan: 0 auto (block counter unless several_slots), 1 fixed deal, 2 items from a block counter
    int scan_blocks = 0;          // pruned scan, blocks per CU: 0 auto (one for small shards when several_slots, else two), 1, 2
    int sample_stride = 0;        // deep-K scans (k > 32): tiles the sample pass skips between two it scores; 0 = library policy
    int run_thresholds = 0;       // deep-K scan (64 < k <= 128): 0 / 1 thresholds tighten during the launch, 2 they stay as the sample pass left them
    int cells_lists = 0;          // pruned scan, who lists a cell's queries: 0 auto, 1 knn_cells_match_kernel, 2 the scan's own waves
    FilterWorkspace ws[KNN_SLOTS];
    // The slots' big scan kernels are chained through this event: two of them sharing the CUs run
    // 15 % slower than back to back; only the small preparation kernels are meant to overlap.
    hipEvent_t scan_done = nullptr;
    bool scan_recorded = false;
};

// ---- cell-pruned form of the filter (knn_cells.hip) -----------------------------------------
#define KNN_CELLS_AUTO_MAX_K 25   // library policy: cell-sorted layouts for resident indexes up to this dimension (`cells` = 1: up to 32)
#define KNN_CELL_BATCH 1024   // queries per pass: their B operands + thresholds sit in 36 KiB of LDS (68 KiB for 16 < k <= 32)
#ifdef __cplusplus
#include <vector>
// Cell codes + counts of the shard (cuts from the strided host sample of the build).  *out stays null when the
// shard does not suit; else *code_out / *fill_out (device; the caller frees them) feed knn_cells_place_rows.
// geom != null: the cell-range shard `rank` of that global grid (no cuts of its own; *out stays null — with *bad_rows_out
// set — when rows fall outside the rank's cell range).
hipError_t knn_cells_build(CellIndex **out, int k, long long n, const float *r_dev, const std::vector<float> &samp,
                           long long samples, hipStream_t s, long long *ntiles_out, unsigned **code_out,
                           unsigned **fill_out, bool one_pass = false, const ShardGeom *geom = nullptr, int rank = 0,
                           unsigned *bad_rows_out = nullptr, bool fast = false, bool defer_scatter = false);
// The fast build in stages (knn_cells_build with defer_scatter: everything allocated and uploaded, nothing scattered yet).
hipError_t knn_cells_fast_scatter(CellIndex &c, int k, const float *r_dev, long long row0, long long row1, hipStream_t s);
hipError_t knn_cells_fast_finish(CellIndex &c, unsigned *counts, hipStream_t s);
#endif
// Sizes of one scan launch of the cell-pruned path (knn_cells.hip; host arithmetic only).
struct CellScanPlan {
    unsigned blocks = 0, waves = 0, nlists = 0, slice = 0, ovf_base = 0, ovf_cap = 0;
    size_t lds_bytes = 0;
};
CellScanPlan knn_cells_scan_plan(int num_cu, int blocks_per_cu, unsigned nitems, unsigned rec_cap, int m_padded,
                                 bool self_lists = false, int kt = 1, bool centred = false);
bool knn_cells_lists_policy(unsigned ncells, bool several_slots);
hipError_t knn_cells_place_rows(FilterState &st, const float *r_dev, const unsigned *code, unsigned *fill, unsigned *out,
                                unsigned ocap, hipStream_t s);
void knn_cells_free(CellIndex *&c);
hipError_t knn_cells_recentre(FilterState &st, const float *r, hipStream_t s);
bool knn_cells_sample_is_clustered(const float *samp, long long samples, int k, float sigma);
hipError_t knn_cells_maybe_recentre(FilterState &st, const float *r, const float *samp, long long samples, hipStream_t s);
extern std::atomic<int> g_knn_cells_centre;
extern std::atomic<long long> g_knn_cells_centred_builds;
void knn_cells_workspace_free(FilterWorkspace &w);
// One batch of <= KNN_CELL_BATCH queries already prepared by the filter's query-fragment kernel: seed, thresholds,
// match, scan (records in w, as the full scan leaves them).  Asynchronous.
// init_keys: the batch's keys are set to (+INF, 0) by the first kernel of the chain.
hipError_t knn_cells_query(FilterState &st, FilterWorkspace &w, int m, const float *q_dev, const float *r_dev, long long base,
                           u64 *keys, int num_cu, bool timed, hipStream_t s, bool init_keys, int *out_idx = nullptr);

// Builds the filter layouts for refs[0..n) (device, AoS).  Synchronous.  Leaves st.usable false
// (and returns hipSuccess) when the data rules the filter out.
// pooled device memory for the CURRENT device (knn_api.cpp); knn_dev_free waits for the device first
hipError_t knn_dev_alloc(void **p, size_t bytes);
hipError_t knn_dev_free(void *p);
// bracket a run of knn_dev_free calls on this thread with ONE device-wide wait (current device)
void knn_dev_free_begin_synced();
void knn_dev_free_end_synced();

// ---- uniform-grid index for k <= 4 (knn_grid.hip) -------------------------------------------
struct GridState;
// *out stays null (hipSuccess) when the data rules the grid out.  Synchronous.
hipError_t knn_grid_build(GridState **out, int k, long long n, const float *r_dev, hipStream_t stream);
void knn_grid_free(GridState *&gs);
// Asynchronous; *gate_out = device word that is != 0 afterwards iff some query left the grid search
// unfinished (the caller queues the gated brute-force scan behind it).
hipError_t knn_grid_query(const GridState *gs, int slot, int m, const float *q_dev, long long base, u64 *keys_dev,
                          const unsigned **gate_out, hipStream_t stream);
void knn_grid_info(const GridState *gs, long long info[4]);

// ---- RCCL exchange step (knn_rccl.cpp; librccl is dlopen'ed at first use) -------------------
#ifdef __cplusplus
#include <string>
int knn_rccl_available(std::string *why);
int knn_rccl_version();
int knn_rccl_comm_sets();   // communicator sets this process has created (0 or 1)
int knn_rccl_allreduce_min(int ndev, const int *devices, u64 *const *keys, int m, const hipStream_t *streams,
                           std::string &err, u64 *const *recv = nullptr);
#endif

// want_cells != 0: also sort the layout into cells (k <= 16, large shards; see CellIndex).
// geom != null (cell-range shard `rank`): centre, scale and cuts are the global grid's; the layout is always cell-sorted.
hipError_t knn_filter_build(FilterState &st, int k, long long n, const float *r_dev, hipStream_t stream,
                            int want_cells = 0, const ShardGeom *geom = nullptr, int rank = 0, unsigned *bad_rows_out = nullptr);
// The global grid of a cell-range sharded set from a sample of it (host rows, samples x k): false when the set does not suit
// (k > 16, too few rows per rank for a cell-sorted layout, a degenerate or non-finite sample).
bool knn_geom_cells(ShardGeom &g, int k, long long n_global, int nranks, const float *sample, long long samples, int seed_tiles);   // (the grid part of it)
bool knn_geom_from_sample(ShardGeom &g, int k, long long n_global, int nranks, const float *sample, long long samples,
                          int seed_tiles);
// owner[i] = rank whose cell range holds rows[i] (device arrays).
hipError_t knn_geom_assign_launch(const ShardGeom &g, const float *rows_dev, long long n, int *owner_dev, hipStream_t s);
// This rank's part of the seed layer, written into the whole-layer buffer `layer_dev` (device; the caller all-gathers the parts).
hipError_t knn_cells_seed_export(const FilterState &st, int rank, unsigned char *layer_dev, hipStream_t s);
// != 0 when gids[0 .. n) is not strictly ascending (device array).  Synchronous.
hipError_t knn_gids_check(const unsigned *gids_dev, long long n, unsigned *bad_out, hipStream_t s);
// Host rows -> device rows (r_dev, n x k floats) + filter layouts, chunk by chunk under the copy.  Synchronous.
hipError_t knn_filter_build_from_host(FilterState &st, int k, long long n, float *r_dev, const float *r_host,
                                      hipStream_t copy, hipStream_t compute);
// Host rows -> device rows + CELL-SORTED layouts, the fast build's bucket pass under the copy.  Always leaves the rows on the
// device; st.usable says whether the layouts stand (else the caller builds from the resident rows).  Synchronous.
hipError_t knn_filter_build_cells_from_host(FilterState &st, int k, long long n, float *r_dev, const float *r_host,
                                            hipStream_t copy, hipStream_t compute);
void knn_filter_free(FilterState &st);
// Asynchronous on `stream`: sample pre-pass + MFMA filter + exact re-rank + gated exact fallback.
// init_keys: the keys are written from scratch ((+INF, 0) first) instead of min-folded into what they hold.
// out_idx (nullable): the int32 indices of the batch as well (no separate unpack launch on the cell-pruned path).
hipError_t knn_filter_query(FilterState &st, int slot, int m, const float *q_dev, const float *r_dev,
                            long long base, u64 *keys_dev, int num_cu, hipStream_t stream,
                            hipEvent_t ev_begin, hipEvent_t ev_end, bool init_keys = false, int *out_idx = nullptr);
// Test hook: raw filter scores S[m][n] (row-major) and the per-query thresholds for a query
// batch, plus {sigma, eta, rho, amax, bmax}.  Synchronous.
hipError_t knn_filter_debug(FilterState &st, int m, const float *q_dev, const float *r_dev,
                            float *scores_dev, float *thr_out_dev, float *qnorm_out_dev,
                            double consts_host[8], hipStream_t stream);
