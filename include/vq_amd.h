/*
 * vq_amd.h -- C ABI of the MI355X-native hot path of PARC-projects/video-query-algorithms.
 *
 * The reference has no FFI of its own: its hot path sits behind two Python-level seams
 * (SURVEY.md 8(b)).  This header is the boundary a maintainer binds underneath those seams
 * (ctypes stub in INTEGRATION.md).  Every entry point names the reference code it replaces
 * (paths relative to the reference checkout).
 *
 * Conventions
 *   - every function returns VQ_OK (0) or a negative VQ_E_* code; vq_last_error() returns the
 *     message of the last failure on the calling thread.  No C++ exception crosses the ABI.
 *   - handles are opaque, bound to one HIP device, own their device memory, and are internally
 *     locked (the reference's broker re-enters from timer threads, broker.py:91-92).
 *   - "host" pointers are ordinary process memory owned by the caller; "dev" pointers are HIP
 *     device addresses on the handle's device.  No torch types anywhere.
 *   - all kernels are launched on the handle's stream (default: the null stream); set it with
 *     vq_*_set_stream(handle, hipStream_t) to interoperate with a framework's stream.
 *
 * Environment switches -- ALL of them (a handle reads its switches when it is created).  None changes a result bit except where
 * it says so; the variants that rounds 1-4 kept compiled in beside the product kernels (the two-kernel batched scan, the
 * two-launch and the square-tile TV-L1 forms, the (p,q,c)-ordered RGB stem, captured forwards) were removed in round 5.
 *   the library
 *     VQ_DEVICE_POOL_GB=<n>        device blocks of closed TSN extractors kept for the next one of the same shape (default 40, 0 = off)
 *     VQ_TSN_SPLIT=<n> | a,b,..    sub-batches of a forward on separate HIP streams (default 2; 1 = one stream)
 *     VQ_SCAN_LEAN=0|1             force the register-lean / register-resident instantiation of the row-major scan (default: by database shape)
 *     VQ_TSN_AUTOTUNE=0|1          0: batch sizes without a tiling table run the occupancy heuristic (nothing is ever timed); 1: the tables
 *                                  shipped beside the library are left out and every size is timed in its first forward; unset: shipped
 *                                  tables, and a timing sweep only for sizes further than 1.6x from every table
 *     VQ_TSN_DEFAULT_TILES=0       leave the shipped tables (tsn/default_tiles.json) out without forcing a sweep elsewhere
 *     VQ_TSN_TILE=BMxBN[xBK[xP]]   one tiling for every direct convolution (tests: every tiling gives the same bits)
 *     VQ_TSN_GROUP=0, VQ_TSN_GROUP_POOL=0   a launch per layer instead of one per graph level / a launch per pooling layer (tests: same bits)
 *     VQ_TSN_SPLITK=0              no split over K for the 7x7-map layers (another summation order: agrees to rounding, tested)
 *     VQ_TSN_POISON=1              NaN-fill every activation slot before a forward (debugging: finds a kernel that leaves output unwritten)
 *     VQ_FLOW_FAST=1               hardware reciprocal / square root in TV-L1's inner iterations: OUTSIDE the tested tolerances (see vq_flow_create)
 *     VQ_JPEG_HOST_HUFFMAN=0|1, VQ_JPEG_DEVICE_MIN_STREAMS=<n>   force the device / host entropy decoder, or move the batch-size threshold (default 2 048 streams); same pixels
 *     VQ_JPEG_STAMPS=1, VQ_JPEG_HOST_STAMPS=1   print the decoder waves' cycle stamps / the host phases of a call (diagnostics)
 *     VQ_RCCL_LIB=<path>           the RCCL the Comm group loads with dlopen (default: the one already mapped, else librccl.so)
 *   the Python package
 *     VQ_AMD_LIB=<path>            libvqamd.so to load (default: next to the package)
 *     VQ_TSN_WINOGRAD=0            every 3x3 layer in direct form (other rounding: both forms are tested against the fp64 oracle)
 *     VQ_TSN_WINO16=0              Winograd layers on maps of at most 14 x 14 without the second filter layout: units of 32 tiles only (default: the
 *                                  tiling table picks 16 or 32 tiles per launch; same bits)
 *     VQ_TUNE_CACHE=<dir>|0, VQ_WEIGHT_CACHE=<dir>|0   where the tiling tables / packed weights are kept between processes (0 = nowhere)
 *     VQ_DIST_BACKEND=gloo|nccl    torch.distributed backend of the N > 1 entry points (default nccl = RCCL; gloo: CPU tests, one-card rehearsals)
 *     VQ_CLI_TRACE=1               calcSig_wOF.py prints phase stamps; VQ_CLI_GROUP_CLIPS=<n>: clips per flush group (default 16 batches per rank)
 *     VQ_INGEST_PRIORITY=low|normal|high   queue priority of the streams the frame-ingest kernels run on (default normal; no priority changed the end-to-end rate: profiles/README.md)
 *     VQ_NO_TORCH=0|1              one-rank calcSig_wOF.py runs WITHOUT importing torch (set to 1 by main() before the library is loaded: device buffers
 *                                  and streams from vq_dev_malloc / vq_stream_*, tsn/devmem.py); 0 keeps torch; same bytes either way
 *     VQ_FANOUT_*                  set BY fanout.py for the per-GPU children it starts (rank, world, device, workers): not for users
 *   the build
 *     VQ_EXTRA_HIPCC_FLAGS         extra hipcc flags for build.py (experiments; empty for the product)
 */
#ifndef VQ_AMD_H
#define VQ_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 6 (round 4): + vq_db_set_layout / vq_db_layout (block tiled in place; the mirrored copy of version 5 is gone), vq_db_read_rows,
 * vq_db_read_scores_at, vq_db_ne_devptr, vq_format_feature_rows, vq_jpeg_decode_path_list; vq_input_desc gained s2d_order.
 * 7 (round 5): + vq_tsn_set_profile_split; the layer tiling tables know the pooled-input kernel (pipelined = 3).
 * 8 (round 5): + vq_resize_crop_planes (the ten grey planes of a batch of flow stacks resized / cropped in one launch).
 * 9 (round 5): + vq_jpeg_crops; vq_jpeg_decode*(color | 2) stops at the component planes.
 * 10 (round 5): + vq_dev_malloc / vq_dev_free / vq_stream_create / vq_stream_destroy / vq_stream_synchronize / vq_dev_read.
 * 11 (round 6): + vq_tsn_tile_tables / vq_tsn_get_tiles / vq_tsn_set_tiles / vq_tsn_tune / vq_tsn_set_split / vq_device_pool_trim; VQ_OP_CONV_WINOGRAD16 (a tiling table is keyed by (batch size, timed side by side on
 *      the sub-batch streams | alone)); vq_db_query_round / vq_db_round_layout / vq_host_alloc / vq_host_free (a query round in one call); vq_db_loss_surface;
 *      vq_stream_create_priority. */
#define VQ_ABI_VERSION 11

enum {
    VQ_OK = 0,
    VQ_E_INVALID = -1,      /* bad argument / shape mismatch            */
    VQ_E_HIP = -2,          /* a HIP runtime call failed                 */
    VQ_E_NOMEM = -3,        /* device or host allocation failed          */
    VQ_E_STATE = -4,        /* call order violated (e.g. scan before query) */
    VQ_E_UNSUPPORTED = -5   /* shape outside what the kernels were built for */
};

enum { VQ_F32 = 0, VQ_F64 = 1 };
enum { VQ_LAYOUT_ROWS = 0, VQ_LAYOUT_TILED = 1 };

const char* vq_last_error(void);
int vq_abi_version(void);
int vq_device_count(int* count);
/* Pure HIP-event timing helpers for bench.py (events recorded on the given stream). */
int vq_timer_create(void** timer);
int vq_timer_start(void* timer, void* stream);
int vq_timer_stop(void* timer, void* stream);
int vq_timer_elapsed_ms(void* timer, float* ms);       /* synchronises on the stop event */
int vq_timer_destroy(void* timer);

/* Feature rows as the text the reference's writer produces (src/features_GPU_compute/calcSig_wOF.py:128-133): per clip
 * "<clip number>,<v0>,...,<vD-1>\n" with every value printed like str(numpy.float64) -- number_format 0: Python's float repr
 * (shortest round-trip digits; numpy >= 1.14), 1: '%.12g' with ".0" kept for integral values (numpy < 1.14); the reference
 * ships feature files of both kinds.  out must hold n_rows * (dim * 26 + 22) bytes; *written = bytes produced.  Host only. */
int vq_format_feature_rows(const double* feats, int64_t n_rows, int32_t dim, const int64_t* clip_numbers, int32_t number_format,
                           char* out, int64_t cap, int64_t* written);

/* ------------------------------------------------------------------------------------------
 * Hot path B: feature database, similarity scan, scoring, selection
 * ------------------------------------------------------------------------------------------ */
typedef struct vq_db vq_db;

/* A resident, row-major [N][S][E][D] feature block (N clips, S streams, E ensemble members
 * ("splits"), D = feature length).  Replaces the nested dict the reference builds from one JSON
 * GET per query in Ticket._get_candidate_features (src/models/ticket.py:358-382).
 * dtype VQ_F32 (BASELINE configs) or VQ_F64 (exact values of the shipped CSVs). */
int vq_db_create(int64_t n, int32_t S, int32_t E, int32_t D, int32_t dtype, int32_t device, vq_db** out);
int vq_db_destroy(vq_db* db);
int vq_db_set_stream(vq_db* db, void* hip_stream);
int vq_db_shape(vq_db* db, int64_t* n, int32_t* S, int32_t* E, int32_t* D, int32_t* dtype);

/* Fill rows [row0,row0+nrows) from host memory laid out [nrows][S][E][D]. */
int vq_db_upload(vq_db* db, int64_t row0, int64_t nrows, const void* feats_host);
/* Use caller-owned device memory ([N][S][E][D], e.g. the output of an RCCL all-gather of
 * per-GPU feature blocks) instead of the handle's own allocation.  The memory must outlive db. */
int vq_db_adopt_device(vq_db* db, void* feats_dev);
/* How the handle's own block is laid out.  VQ_LAYOUT_ROWS: [N][S][E][D] (the default; what vq_db_upload, vq_db_adopt_device and
 * vq_db_feats_devptr speak).  VQ_LAYOUT_TILED: [tile of 16 clips][S*E][D/4][clip][4] -- the order the 16-query pass wants, in which
 * every load instruction of a wave (one-query scan AND 16-query pass) takes 1 KB of contiguous memory.  The conversion is IN PLACE
 * (a tile's rows and its tiled form cover the same bytes; no second copy of the database, 256 MB of scratch while it runs) and
 * costs one sweep of the block: call it once after loading, off the query path.  Tiled needs fp32, D = 1024, S <= 2, E <= 5
 * (VQ_E_UNSUPPORTED otherwise) and a block that is the library's alone (VQ_E_STATE after vq_db_adopt_device / vq_db_feats_devptr).
 * Every other entry point works on either layout; results of the two layouts agree to rounding (<= 1e-12: the k order of a
 * dot differs), each is bit-reproducible in itself.  vq_db_upload into a tiled database deals the rows into their tiles. */
int vq_db_set_layout(vq_db* db, int32_t layout);
int vq_db_layout(vq_db* db, int32_t* layout);
/* Optional [N][S][E] presence mask (1 = this clip has this split); NULL restores "dense".
 * Mirrors clips that lack a split in ticket.py:146-160 (n_e = number of splits present). */
int vq_db_set_present(vq_db* db, const uint8_t* present_host);
/* Synthetic rows generated on the device from a counter-based hash (too big to ship):
 * value(row,s,e,d) = u24(hash(seed, flat index of global row)) * 2^-24 * scales[s]. */
int vq_db_generate(vq_db* db, uint64_t seed, int64_t global_row0, const float* scales_host);
int vq_db_feats_devptr(vq_db* db, void** dev_ptr);

/* Query vectors t[S][E][D] (fp64): the reference's target_features,
 * TargetClip.scaled_ref_clip_features / _scale_feature (src/models/target_clip.py:137-143,311-313). */
int vq_db_set_query(vq_db* db, const double* t_host);
/* Device-side restatement of _scale_feature for a ref clip that lives in the DB:
 * t[s][e] = r / (r . r) with r = row `row` (fp64 arithmetic).  t_out_host may be NULL. */
int vq_db_set_query_from_row(vq_db* db, int64_t row, double* t_out_host);

/* Target bootstrapping ("dynamic target adjustment", src/models/target_clip.py:161-261): the closed-form new query
 * vector of one (stream, split) from its user-validated clips.  With X = the n_valid validated matches and Y = the
 * n_invalid validated non-matches (rows of `dim` values):
 *   n_invalid == 0 (or mu == 0):  w = X^T (X X^T)^-1 1                         (_bootstrap_valid_matches, :192-197)
 *   otherwise:                    the minimiser built at :245-260 from M = I + (mu / tr(Y Y^T)) Y^T Y
 *                                 (_bootstrap_valid_plus_invalid).
 * Computed on the device in fp64 from the Gram matrix of the rows (Woodbury form: two solves of size <= n_valid +
 * n_invalid instead of the reference's dim x dim inverses).  n_valid + n_invalid <= 256 per problem.
 * rows_host: the problems back to back, each its valid rows then its invalid rows, dtype VQ_F32 / VQ_F64.
 * targets_host: [n_problems][dim] fp64.  VQ_E_INVALID if a system is singular (numpy.linalg.inv raises there). */
int vq_bootstrap_targets(const void* rows_host, int32_t dtype, int32_t n_problems, const int32_t* n_valid,
                         const int32_t* n_invalid, int32_t dim, double mu, int32_t device, double* targets_host);
/* Same for clips that live in the DB: problem (s, e) uses rows valid_rows / invalid_rows of stream s, split e (every
 * listed clip must hold every stream and split).  targets_host [S][E][D] may be NULL; set_query != 0 installs the
 * targets as the query vectors (as vq_db_set_query would). */
int vq_db_bootstrap_target(vq_db* db, const int64_t* valid_rows, int32_t n_valid, const int64_t* invalid_rows,
                           int32_t n_invalid, double mu, double* targets_host, int32_t set_query);

/* One pass over the DB: sim[c][s][e] = t[s][e] . x[c][s][e]; avg[c][s] = sum_e sim / n_e;
 * if w != NULL also score[c] = 1 - sqrt(sum_s (w_s (1 - avg))^2 / sum_s w_s^2).
 * Replaces Ticket.compute_similarities (ticket.py:120-163) + compute_scores (ticket.py:165-180).
 * keep_sims != 0 additionally keeps the per-split dot products for vq_db_read_similarities. */
int vq_db_scan(vq_db* db, const double* w_host, int32_t keep_sims);
/* score[c] from the cached avg[N][S] (no DB read): Ticket.compute_scores, ticket.py:165-180. */
/* Batched scan: n_queries (<= 16) queries in ONE pass over the database -- the all-pairs form of the scan for a broker
 * that holds several tickets.  Per (query, clip) the arithmetic of vq_db_scan with weights (ticket.py:120-180); the dots
 * run on the fp64 matrix cores (v_mfma_f64_16x16x4: tiles of 16 clips x 16 queries), whose accumulation order differs
 * from the single-query scan's, so the scores agree with it to rounding (<= 1e-12), not bit for bit.  The database is
 * read once (slice by slice, the slice's query vectors in LDS) and the pass is HBM-bound: 16 queries cost what one does.
 * t_host [n_queries][S][E][D] fp64, w_host [n_queries][S] fp64, scores_host [n_queries][N] fp64 (may be NULL: use
 * vq_db_batch_scores_devptr).  Does not touch the state of the single-query path (query, avg, scores).
 * Needs D in {256, 512, 768, 1024}.
 * On a row-major database a load instruction of the pass takes 64 bytes of each of 16 clips; on a TILED one
 * (vq_db_set_layout) whole 128-byte lines: 8 % less time per pass, same scores bit for bit. */
int vq_db_scan_batch(vq_db* db, int32_t n_queries, const double* t_host, const double* w_host, double* scores_host);
int vq_db_batch_scores_devptr(vq_db* db, void** dev_ptr /* double [n_queries][N] */, int32_t* n_queries);
int vq_db_rescore(vq_db* db, const double* w_host);
/* Copy results to the host.  Any pointer may be NULL.  avg [N][S], n_e [N][S], sims [N][S][E]. */
int vq_db_read_similarities(vq_db* db, double* avg_host, int32_t* n_e_host, double* sims_host);
int vq_db_read_scores(vq_db* db, double* scores_host);
/* out[l] = score of row rows[l] (a sharded selection asks each rank for the score of its own best near miss). */
int vq_db_read_scores_at(vq_db* db, const int64_t* rows_host, int32_t L, double* out_host);
/* Device addresses of the result arrays (for an RCCL all-gather of score slices). */
int vq_db_scores_devptr(vq_db* db, void** dev_ptr);
int vq_db_avg_devptr(vq_db* db, void** dev_ptr);
int vq_db_ne_devptr(vq_db* db, void** dev_ptr /* int32 [N][S] */);
/* Rows of the resident block back to the host: out [L][S][E][D] in the database's dtype.  The sharded B seam uses it to
 * send the handful of user-validated clips of a round (src/models/target_clip.py:114-135) to the rank that solves the
 * bootstrapping problems; features never travel otherwise. */
int vq_db_read_rows(vq_db* db, const int64_t* rows_host, int32_t L, void* out_host);
/* Replace the cached avg[N][S]/n_e (e.g. after gathering slices from other ranks). */
int vq_db_write_avg(vq_db* db, const double* avg_host, const int32_t* n_e_host);

/* out[g][l] = score of row rows[l] under weights w_grid[g][0..S): the 40 rescorings of
 * Hyperparameter.optimize_weights (src/models/hyperparameter.py:57-58) restricted to the
 * labelled clips that the loss (hyperparameter.py:60-64) actually reads. */
int vq_db_scores_grid(vq_db* db, const double* w_grid_host, int32_t G, const int64_t* rows_host, int32_t L,
                      double* out_host);

/* Hyperparameter.optimize_weights' loss surface (hyperparameter.py:57-64) in ONE launch: for each of the G grid weights (w_grid [G][S]) the
 * scores of the L labelled rows (as vq_db_scores_grid), and over the T grid thresholds
 *     out[g][t] = 0.5 th[t] + sum over the labels IN ORDER of (heaviside(score - th[t], 1) - y) * (score - th[t]) * (1 + y * ballast)
 * with the reference's elementwise operations in its order (fp64, no contraction): every cell sees the sequence of additions of the
 * reference's loop over `inferred_validations` (the caller divides by L).  labels: 1.0 / 0.0 per row.  L <= 4096. */
int vq_db_loss_surface(vq_db* db, const double* w_grid_host, int32_t G, const int64_t* rows_host, const double* labels_host, int32_t L,
                       const double* th_grid_host, int32_t T, double ballast, double* out_host);

/* Order-preserving partition of the scores, Ticket.select_clips_to_review (ticket.py:325-340):
 * match = {v >= threshold}, near = {lower <= v < threshold}, near_argmax = first row of the
 * largest near score (-1 if none).  Lists stay on the device until vq_db_select_fetch. */
int vq_db_select(vq_db* db, double threshold, double lower, int64_t* n_match, int64_t* n_near,
                 int64_t* near_argmax);
int vq_db_select_fetch(vq_db* db, int64_t* match_rows_host, int64_t cap_match, int64_t* near_rows_host,
                       int64_t cap_near);
/* vq_db_select + vq_db_select_fetch under ONE lock of the handle (broker threads share handles: a top-k or another
 * selection can otherwise slip in between the two calls).  The two host buffers must hold cap_match / cap_near rows
 * (N each is always enough); VQ_E_INVALID with the counts filled in if one is too small. */
int vq_db_select_rows(vq_db* db, double threshold, double lower, int64_t* match_rows_host, int64_t cap_match,
                      int64_t* near_rows_host, int64_t cap_near, int64_t* n_match, int64_t* n_near, int64_t* near_argmax);
/* One query round on a resident database as ONE locked call with ONE synchronisation and one copy back: what
 * Ticket.compute_similarities + compute_scores + select_clips_to_review (ticket.py:120-180,311-356; called in that order by
 * compute_matches.py:58-89) are per query.  ``block`` is page-locked host memory (vq_host_alloc) laid out as vq_db_round_layout says
 * (byte offsets, every piece 64-byte aligned):
 *     off[0] query t [S][E][D] f64 (in)      off[1] weights [8] f64 (in)
 *     off[2] avg [N][S] f64                  off[3] n_e [N][S] i32              off[4] scores [N] f64
 *     off[5] result i64 [4] = n_match, n_near, near_argmax (-1: none), 0
 *     off[6] the first off[9] match rows     off[7] the first off[9] near rows   (both in database order)
 *     off[8] bytes of the block              off[9] rows in each prefix list (1 024)
 * flags: 1 = take the query from the block and scan (ticket.py:120-163; else the similarities the handle holds are used),
 *        2 = scores under the block's weights (ticket.py:165-180; fused into the scan when both are asked for),
 *        4 = the order-preserving partition of vq_db_select under (threshold, lower).
 * What was computed is copied back (from avg on after a scan, from scores on otherwise).  A list longer than its prefix is fetched in
 * full with vq_db_select_fetch.  The handle's state afterwards is that of the separate calls (vq_db_set_query, vq_db_scan, vq_db_rescore,
 * vq_db_select), whose results this call equals bit for bit (tests/test_ticket_gpu.py); per-query slot restrictions stay with
 * vq_db_set_present. */
int vq_db_round_layout(vq_db* db, int64_t off[10]);
int vq_db_query_round(vq_db* db, void* block, int64_t block_bytes, int32_t flags, double threshold, double lower);
/* Page-locked host memory for such blocks (device copies to and from it are asynchronous and need no staging). */
int vq_host_alloc(void** ptr, int64_t bytes);
int vq_host_free(void* ptr);
/* Rows of the k largest scores, descending, ties by ascending row (the stable sort of the final
 * report, ticket.py:266).  *k_out = min(k, number of non-NaN scores). */
int vq_db_topk(vq_db* db, int64_t k, int64_t* rows_host, double* vals_host, int64_t* k_out);
/* min over rows[] of score (init 1), Ticket.lowest_scoring_user_match (ticket.py:301-309). */
int vq_db_min_score(vq_db* db, const int64_t* rows_host, int32_t L, double* min_out);

/* ------------------------------------------------------------------------------------------
 * Hot path A: TSN BN-Inception forward + segment consensus
 * ------------------------------------------------------------------------------------------ */
typedef struct vq_tsn vq_tsn;

/* Frame ingest (the step in front of vq_tsn_forward): n decoded frames [n][h][w][c] uint8 (host, or device if
 * frames_on_device) -> cv2.resize(frame, (resize_w, resize_h)), INTER_LINEAR -> over-sample crop 0 (top-left
 * crop x crop), written as channels [dst_channel0, dst_channel0 + c) of the device buffer crops_dev
 * [n][crop][crop][dst_channels] (so the 10 grey flow frames of a stack interleave into one 10-channel crop).
 * Replaces the resize/over-sample half of CaffeNet.predict_single_frame / predict_single_flow_stack(...,
 * frame_size=(340,256)) at calcSig_wOF.py:94,111.  rule: VQ_RESIZE_CV2_FIXED = the rule of the reference's dependency,
 * OpenCV's uint8 path (float sample positions, 11-bit fixed-point weights, integer passes; same bytes as
 * tsn/frames.py:resize_cv2_fixed and oracle/frames_oracle.py); VQ_RESIZE_EXACT = the same sampling grid with exact fp64
 * weights, round half to even (tsn/frames.py:resize_exact).  A frame that already has the size is copied.  Both are
 * restated without cv2 at hand: parity unpinned. */
#define VQ_RESIZE_CV2_FIXED 0
#define VQ_RESIZE_EXACT 1
int vq_resize_crop(const uint8_t* frames, int32_t frames_on_device, int32_t n, int32_t h, int32_t w, int32_t c,
                   int32_t resize_w, int32_t resize_h, int32_t crop, int32_t rule, uint8_t* crops_dev, int32_t dst_channels,
                   int32_t dst_channel0, int32_t device, void* hip_stream);

/* The same for the ten grey planes of a batch of flow stacks in ONE launch: plane p of frame i at planes_dev + p * plane_stride + i * h * w
 * (what vq_jpeg_decode leaves on the device when the files are handed over plane-major), written as the interleaved crops
 * [n][crop][crop][c].  c must be 10 (predict_single_flow_stack's stack of 5 x/y pairs, calcSig_wOF.py:98-111); crop even.  Same bytes as c
 * calls of vq_resize_crop (tested); a thread computes the taps of its two pixels once for all planes and stores whole words. */
int vq_resize_crop_planes(const uint8_t* planes_dev, int32_t n, int32_t h, int32_t w, int32_t c, int64_t plane_stride, int32_t resize_w,
                          int32_t resize_h, int32_t crop, int32_t rule, uint8_t* crops_dev, int32_t device, void* hip_stream);

/* Device plumbing for a host that has no tensor library loaded (the single-GPU command line runs without importing torch: tsn/devmem.py):
 * a device buffer, a non-blocking stream, a wait (stream NULL: the default stream), a blocking read-back. */
int vq_dev_malloc(void** ptr, int64_t bytes, int32_t device);
int vq_dev_free(void* ptr, int32_t device);
int vq_stream_create(void** hip_stream, int32_t device);
/* The same with a queue priority: -1 the lowest the device offers, 0 the middle, +1 the highest.  The frame-ingest kernels in front of the
 * networks (JPEG entropy decoding, IDCT, pixel and resize passes of LATER batches: calcSig_wOF.py:88-113 decodes inside the loop) run on
 * streams whose priority VQ_INGEST_PRIORITY sets (measured without effect on the networks: profiles/README.md). */
int vq_stream_create_priority(void** hip_stream, int32_t device, int32_t priority);
int vq_stream_destroy(void* hip_stream, int32_t device);
int vq_stream_synchronize(void* hip_stream, int32_t device);
int vq_dev_read(void* host, const void* dev, int64_t bytes, int32_t device);

/* JPEG decode, the step in front of vq_resize_crop: replaces cv2.imread(img_NNNNN.jpg, IMREAD_COLOR) and
 * cv2.imread(flow_{x,y}_NNNNN.jpg, IMREAD_GRAYSCALE) at calcSig_wOF.py:92,105-106 for the baseline JPEG files
 * build_wof_clips.py:46,70-73 writes.  Entropy decoding (ITU-T T.81 F.2.2) runs on host threads, one frame each; the
 * integer "islow" IDCT, libjpeg's fancy h2v2 / h2v1 chroma upsampling and its fixed-point YCbCr -> RGB run on the device for
 * the whole batch -- the same bits libjpeg(-turbo), and therefore cv2.imread, produces (oracle/jpeg_oracle.py, pinned against
 * Pillow's libjpeg-turbo).  n files of ONE size h x w; color != 0: [n][h][w][3] B,G,R; color == 0: [n][h][w] (the Y plane,
 * as cv2's grey read of a colour file).  out_host (optional) receives the frames; *out_dev (optional) is set to the
 * handle's device copy, valid until the next call -- pass it to vq_resize_crop(frames_on_device = 1).
 * Progressive / arithmetic / 12-bit / multi-scan / CMYK files: VQ_E_UNSUPPORTED. */
typedef struct vq_jpeg vq_jpeg;
int vq_jpeg_info(const uint8_t* data, int64_t size, int32_t* h, int32_t* w, int32_t* components);
int vq_jpeg_create(int32_t max_frames, int32_t max_h, int32_t max_w, int32_t device, vq_jpeg** out);
int vq_jpeg_destroy(vq_jpeg* jpeg);
int vq_jpeg_decode(vq_jpeg* jpeg, const uint8_t* const* files, const int64_t* sizes, int32_t n, int32_t color, int32_t h, int32_t w,
                   uint8_t* out_host, uint8_t** out_dev, void* hip_stream);
/* The same on file paths: the library's worker threads read the files (no interpreter in the loop); vq_jpeg_info_file reads
 * a file's frame header. */
int vq_jpeg_decode_files(vq_jpeg* jpeg, const char* const* paths, int32_t n, int32_t color, int32_t h, int32_t w, uint8_t* out_host,
                         uint8_t** out_dev, void* hip_stream);
/* ... and on a path LIST: n paths back to back in one buffer of paths_bytes bytes, each closed by its NUL (a caller under an
 * interpreter builds one bytes object instead of an array of thousands of pointers). */
int vq_jpeg_decode_path_list(vq_jpeg* jpeg, const char* paths, int64_t paths_bytes, int32_t n, int32_t color, int32_t h, int32_t w,
                             uint8_t* out_host, uint8_t** out_dev, void* hip_stream);
int vq_jpeg_info_file(const char* path, int32_t* h, int32_t* w, int32_t* components);
/* Decode + crop 0 without the whole-frame pixel pass, for frames that already have the resize size (what build_wof_clips.py writes: 340 x 256;
 * cv2.resize to the same size is the identity, so the over-sample crop 0 of calcSig_wOF.py:94,111 is the top-left crop x crop pixels).
 * A decode call with (color | 2) stops at the component planes (no out_host, *out_dev = NULL); vq_jpeg_crops then writes
 *   c == 3:  [n][crop][crop][3] B,G,R of the n colour frames of that call,
 *   c == 10: [n / 10][crop][crop][10] of n grey frames handed over plane-major (frame p * (n / 10) + i = plane p of stack i)
 * into crops_dev -- the bytes of vq_jpeg_decode + vq_resize_crop(_planes) (tested). */
int vq_jpeg_crops(vq_jpeg* jpeg, int32_t c, int32_t crop, uint8_t* crops_dev, void* hip_stream);

enum {
    VQ_OP_CONV = 1,
    VQ_OP_MAXPOOL = 2,
    VQ_OP_AVGPOOL = 3,
    VQ_OP_GLOBAL_AVGPOOL = 4,
    /* 3x3 / stride 1 / pad 1 convolution evaluated as Winograd F(2x2,3x3): same result up to fp32 rounding
     * (2.25x fewer multiplies).  w_off then addresses the host-transformed filters U = G g G^T laid out
     * [cin/8][16][cout][8] (position 4 i + j of the 4x4 transform, 8 consecutive input channels innermost). */
    VQ_OP_CONV_WINOGRAD = 5,
    /* The same, able to run in units of 16 tiles on v_mfma_f32_16x16x4_f32 as well (layers on maps of at most 14 x 14, some of whose
     * launches have too few 32-tile units for the chip: the tiling table says which form a launch takes -- tile (64, 32|64, 16, 2)
     * instead of (128, 32|64, 8, 2)).  w_off addresses BOTH layouts, one behind the other: the one above, then [cin/16][16][cout][16],
     * slot p of the innermost 16 = input channel {0,2,8,10, 4,6,12,14, 1,3,9,11, 5,7,13,15}[p] of the group (csrc/vq_wino.hip).
     * cin % 16 == 0.  Every form gives the same bits. */
    VQ_OP_CONV_WINOGRAD16 = 6
};

/* One executed layer of the frozen network
 * (src/features_GPU_compute/models/ucf101/tsn_bn_inception_{rgb,flow}_deploy.prototxt).
 * Convolution + frozen BN + ReLU are one op (BN folded into weights/bias by the host);
 * Concat is expressed by writing at a channel offset of the destination tensor. */
typedef struct vq_layer_desc {
    int32_t op;
    int32_t src, dst;            /* tensor slots                                             */
    int32_t src_coff, dst_coff;  /* first channel read / written inside the slot             */
    int32_t cin, cout;           /* channels read / written                                  */
    int32_t k, stride, pad;
    int32_t relu;
    int32_t ceil_mode;           /* pooling output size: Caffe ceil rule                     */
    int32_t has_bias;            /* AVGPOOL: add bias[b_off..] after the division, then ReLU if relu (finishes a
                                    1x1 projection that was commuted with the pool); conv: always biased     */
    int32_t seg_first, seg_count;/* conv with seg_count > 0: its cout columns are split over seg_count
                                    destinations (segments[seg_first ..]); dst/dst_coff/relu are then ignored */
    int32_t pre_pool_k;          /* VQ_OP_CONV, 1x1/1/0 only: 3 = the convolution reads max over a 3x3 window (stride     */
    int32_t pre_pool_stride;     /* pre_pool_stride, no padding, Caffe ceil rule) of src, i.e. a MAX Pooling layer and the
                                    1x1 convolution behind it (pool1/3x3_s2 -> conv2/3x3_reduce, prototxt :44-75) in one
                                    launch, the pooled tensor never written; 0 = plain convolution                        */
    int64_t w_off;               /* floats into the blob: weights [cout][k][k][cin] (OHWI)   */
    int64_t b_off;               /* floats into the blob: bias [cout]                        */
} vq_layer_desc;

/* One destination of a multi-destination convolution: `cout` consecutive output columns (a multiple of 32) go to
 * tensor slot `dst` at channel offset `dst_coff`.  Sibling 1x1 convolutions of an inception block that read the same
 * tensor are executed as one GEMM this way. */
typedef struct vq_conv_segment {
    int32_t cout, dst, dst_coff, relu;
} vq_conv_segment;

typedef struct vq_tensor_desc {
    int32_t h, w, c;             /* NHWC activations, fp32                                    */
} vq_tensor_desc;

/* The crops handed to vq_tsn_forward: h x w x c uint8, and how slot 0 (the first layer's source) is produced from
 * them on the device (value = pixel - mean[channel], zero outside the crop):
 *   s2d_pad < 0   slot 0 = the h x w image, channels padded with zeros to tensors[0].c (a multiple of 4, < c + 4);
 *   s2d_pad >= 0  slot 0 = the space-to-depth(2) image of the crop shifted by s2d_pad:
 *                 slot0[Y][X][(p*2+q)*c + ch] = crop[2Y + p - s2d_pad][2X + q - s2d_pad][ch],  tensors[0].c = 4c.
 *                 A k x k / stride-2 / pad-s2d_pad first convolution is then an exact ceil(k/2) x ceil(k/2) / stride-1 /
 *                 pad-0 convolution over slot 0 whose K dimension has no channel padding (the 7x7x3 stem: K = 192
 *                 instead of 7*7*4 -> 224); its weights must be packed accordingly (zeros for the tap k). */
typedef struct vq_input_desc {
    int32_t h, w, c;
    int32_t s2d_pad;
    int32_t s2d_kernel;          /* with s2d_pad >= 0: the k of the original first convolution */
    int32_t s2d_order;           /* 0: cell channels (p*2+q)*c + ch as above.  1 (7x7 stem): x-major cells, (q*2+p)*c + ch -- the seven
                                  * x-taps of a kernel row are then 14c contiguous floats of slot 0, and the first convolution's weights
                                  * are packed [Cout][ceil(14c/4)][4][4] = W[o][ch][2r+p][t] at [o][s][r][e] with 4s + e = 2ct + cp + ch
                                  * (zero for 2r+p = 7 and behind the run): K = 176 instead of 192 (c = 3), 560 instead of 640 (c = 10) */
} vq_input_desc;

/* `feature_slot` names the 1x1xD tensor that is the feature blob ("global_pool",
 * calcSig_wOF.py:95,112,174-175).  Replaces CaffeNet(proto, weights, device) at
 * calcSig_wOF.py:52,55. */
int vq_tsn_create(const vq_tensor_desc* tensors, int32_t n_tensors, const vq_layer_desc* layers,
                  int32_t n_layers, const vq_conv_segment* segments, int32_t n_segments, const float* blob_host,
                  int64_t blob_floats, const vq_input_desc* input, int32_t feature_slot, int32_t max_crops, int32_t device,
                  vq_tsn** out);
int vq_tsn_destroy(vq_tsn* net);
int vq_tsn_set_stream(vq_tsn* net, void* hip_stream);
/* crops: uint8 NHWC [n_crops][h][w][c] (host, or device if crops_on_device), n_crops = B*T with the
 * T snippets of one clip contiguous (h, w, c of the vq_input_desc).  mean[c] is subtracted per channel (BGR [104,117,123] /
 * flow 128).  Outputs (host, may be NULL): per_snippet [n_crops][D] fp32 = the global_pool blob of
 * each snippet (calcSig_wOF.py:95,112); feat [B][D] fp64 = the segment consensus
 * np.array(frame_features).mean(axis=0) (calcSig_wOF.py:82).
 * Replaces the per-snippet loop of rgbFeatureExtract / flowFeatureExtract (calcSig_wOF.py:88-113). */
int vq_tsn_forward(vq_tsn* net, const uint8_t* crops, int32_t crops_on_device, int32_t n_crops, int32_t T,
                   const float* mean_host, double* feat_host, float* per_snippet_host);
/* Same, outputs left on the device (bench / all-gather path). */
int vq_tsn_feat_devptr(vq_tsn* net, void** feat_dev /* double [B][D] */, void** per_snippet_dev);
/* Copy an activation slot of the last forward to the host ([n_crops][h][w][c] fp32): per-layer parity. */
int vq_tsn_read_tensor(vq_tsn* net, int32_t slot, int32_t n_crops, float* host);
/* Roofline accounting: with depth > 0 every launch of a forward carries a start and a stop HIP event
 * (hipExtLaunchKernelGGL: the dispatch packet's own begin / end timestamps, i.e. the kernel alone, as rocprofv3
 * reports it) from a ring of `depth` event sets -- no host synchronisation and no extra packets inside the forward;
 * profiled forwards run on the handle's stream only (no batch split).
 * vq_tsn_layer_times returns the per-layer device time averaged over the profiled forwards since
 * vq_tsn_set_profile (a launch shared by several layers is split between them in proportion to their matrix-core work) and (optionally) each layer's algorithmic FLOPs for the last batch size (2*MACs with
 * the un-padded channel counts; 0 for pooling).  depth = 0 switches profiling off. */
int vq_tsn_set_profile(vq_tsn* net, int32_t depth);
/* While profiling, only every `every`-th forward carries the events (the start/stop signals cost ~3 us per launch);
 * the forwards in between issue exactly the same launches on the same stream.  Default 1. */
int vq_tsn_set_profile_every(vq_tsn* net, int32_t every);
/* While profiling with every > 1: how the forwards BETWEEN two sampled ones run.  0 (default): like the sampled ones, on the
 * handle's stream only (every forward of the region issues the same launches: what a rocprofv3 comparison wants).  1: as the
 * product runs them when nobody profiles -- the batch split into VQ_TSN_SPLIT sub-batches on separate HIP streams; only the
 * sampled forwards stay on one stream, so that their per-launch durations are each kernel alone on the chip (bench.py's timed
 * region: the product's own mode with a clean kernel sample inside it).  Same bits either way. */
int vq_tsn_set_profile_split(vq_tsn* net, int32_t unsampled_split);
int vq_tsn_layer_times(vq_tsn* net, float* ms, double* flops, int32_t n_layers);
/* The implicit-GEMM tiling (BM, BN, BK, pipelined?) each conv layer runs with at batch size n_crops: autotuned on the first
 * forward of that batch size (every candidate yields the same bits), else the occupancy heuristic.
 * tiles: int32 [n_layers][4], zeros for non-conv layers. */
int vq_tsn_layer_tiles(vq_tsn* net, int32_t n_crops, int32_t* tiles, int32_t n_layers);
/* Install a tiling table (as returned by vq_tsn_layer_tiles) for batch size n_crops instead of autotuning. */
int vq_tsn_set_layer_tiles(vq_tsn* net, int32_t n_crops, const int32_t* tiles, int32_t n_layers);
/* The launch sequence of a forward: layers are levelled by their dependencies and run level by level; the Winograd
 * convolutions of one level (the 3x3 and the first double-3x3 arm of an inception module) share ONE kernel launch.
 * item_of_layer[l] = index of the launch that executes layer l; *n_items = launches per forward (per sub-batch).
 * VQ_TSN_GROUP=0 (read at creation) gives every layer its own launch; results never depend on the grouping. */
int vq_tsn_launch_items(vq_tsn* net, int32_t* item_of_layer, int32_t n_layers, int32_t* n_items);
/* Batch sizes that hold a tiling table (autotuned or installed): *n of them, the first min(*n, cap) in sizes[]. */
int vq_tsn_tuned_sizes(vq_tsn* net, int32_t* sizes, int32_t cap, int32_t* n);
/* A handle holds one tiling table per (batch size, paired): ``paired`` tables belong to the SUB-BATCH sizes of the default forward
 * (calcSig_wOF.py:88-113 runs one net.forward per snippet; here a batch is split into VQ_TSN_SPLIT sub-batches that run side by side on
 * separate HIP streams, and a launch beside its twin prefers other tilings than one that has the chip to itself); un-paired tables
 * belong to forwards that run on one stream.  vq_tsn_tile_tables lists them: *n tables, the first min(*n, cap) in sizes[] / flags[]
 * (bit 0: paired; bit 1: borrowed from a neighbouring size within 1.6x instead of measured or installed -- such a table is never worth
 * persisting).  vq_tsn_get_tiles / vq_tsn_set_tiles read / install one table (layout as vq_tsn_layer_tiles); the older
 * vq_tsn_layer_tiles / vq_tsn_set_layer_tiles address the un-paired table of a size (reading falls back to the paired one).
 * Tables come from three places, in this order: the machine's own tuning cache, the default tables shipped beside the library for the
 * BASELINE shapes on gfx950 (tsn/default_tiles.json), a timing sweep inside the first forward of an unseen size (VQ_TSN_AUTOTUNE=0: the
 * occupancy heuristic instead; =1: sweep even where a shipped table exists).  No table changes a result bit. */
int vq_tsn_tile_tables(vq_tsn* net, int32_t* sizes, int32_t* flags, int32_t cap, int32_t* n);
/* Give the device blocks closed extractors left in the process-wide pool (VQ_DEVICE_POOL_GB) back to the driver now. */
int vq_device_pool_trim(void);
/* Whether a forward of (about) n_crops is cut into sub-batches on separate streams: split = 0: one stream (measured faster at that size:
 * the flow network at 448 crops, tools/make_default_tiles.py), 1: sub-batches, -1: forget the entry.  A forward takes the entry of the
 * nearest size within 1.6x; without one it splits (VQ_TSN_SPLIT).  Same bits either way. */
int vq_tsn_set_split(vq_tsn* net, int32_t n_crops, int32_t split);
/* Run the timing sweep for one table now (what the first forward of an unseen size does by itself; tools/make_default_tiles.py). */
int vq_tsn_tune(vq_tsn* net, int32_t n_crops, int32_t paired);
int vq_tsn_get_tiles(vq_tsn* net, int32_t n_crops, int32_t paired, int32_t* tiles, int32_t n_layers);
int vq_tsn_set_tiles(vq_tsn* net, int32_t n_crops, int32_t paired, const int32_t* tiles, int32_t n_layers);
/* Algorithmic FLOPs (2*MACs of the conv layers) of one crop, for roofline accounting. */
int vq_tsn_flops_per_crop(vq_tsn* net, double* flops);

/* ------------------------------------------------------------------------------------------
 * Frame / flow preparation in front of hot path A: TV-L1 optical flow
 * ------------------------------------------------------------------------------------------
 * Replaces the arithmetic of src/features_GPU_compute/build_wof_clips.py:55-76, which shells out to the third-party
 * binary `extract_warp_gpu -b 20 -t 1 -s 1` (OpenCV CUDA TV-L1; not in the reference tree -> PARITY UNPINNED).  The
 * kernels follow the published algorithm (Zach, Pock & Bischof 2007; IPOL 2013 Algorithm 1) with OpenCV's default
 * parameters, exactly as restated in oracle/tvl1_oracle.py.  A homography per pair can be supplied and is applied to the
 * second frame; vq_flow_good_features + vq_flow_ransac_homography below estimate it the way the binary's flow-match
 * branch does (its SURF matches are not built). */
typedef struct vq_flow vq_flow;
typedef struct vq_tvl1_params {
    float tau, lambda, theta;    /* 0.25, 0.15, 0.3  */
    float epsilon;               /* 0.01: a warp's inner loop stops when the mean squared update <= epsilon^2 */
    float scale_step;            /* 0.8  */
    int32_t nscales, warps, iterations;   /* 5, 5, 300 */
    float bound;                 /* 20: flow images map [-bound, bound] px to [0, 255] (extract_warp_gpu -b 20) */
} vq_tvl1_params;
int vq_tvl1_default_params(vq_tvl1_params* params);
/* A batch workspace for up to max_pairs frame pairs of h x w grey pixels.  params NULL = defaults.
 * Environment, read here: VQ_FLOW_FAST=1 -- hardware reciprocal / square root (1 ulp) instead of IEEE division / sqrtf in the inner
 * iterations (faster; flow images differ from the default's on ~0.4 % of the pixels, without bound where the flow is not determined:
 * outside the tolerances the kernels are tested to, never on by default, never in a reported headline). */
int vq_flow_create(int32_t max_pairs, int32_t h, int32_t w, const vq_tvl1_params* params, int32_t device, vq_flow** out);
int vq_flow_destroy(vq_flow* flow);
/* Pyramid actually used: *n_levels and (h, w) of the first min(*n_levels, cap) levels, finest first. */
int vq_flow_levels(vq_flow* flow, int32_t* n_levels, int32_t* sizes_hw, int32_t cap);
/* Flow from frames0[p] to frames1[p] (uint8 grey [n_pairs][h][w], host or device) for every pair of the batch.
 * homographies_host (optional, [n_pairs][9] row-major fp64): frames1[p] is first warped by it (cv::warpPerspective
 * semantics, bilinear, replicated border).  Outputs, all optional, host: u1 / u2 = dx / dy fp32 [n_pairs][h][w];
 * flow_x / flow_y = the 8-bit images extract_warp_gpu writes; iters [levels][warps][n_pairs] (coarsest level first) =
 * inner iterations each pair ran.  Synchronises on hip_stream before returning. */
int vq_flow_tvl1(vq_flow* flow, const uint8_t* frames0, const uint8_t* frames1, int32_t frames_on_device, int32_t n_pairs,
                 const double* homographies_host, float* u1_host, float* u2_host, uint8_t* flow_x_host, uint8_t* flow_y_host,
                 int32_t* iters_host, void* hip_stream);
/* Device time of the inner loops (the iteration kernels of every level and warp: HIP events around each loop, summed) and the
 * number of iteration-kernel launches of the LAST vq_flow_tvl1 call on this handle -- what bench.py prices the kernel with. */
int vq_flow_last_timing(vq_flow* flow, double* inner_loops_ms, int32_t* iteration_launches);
/* The warped flow of extract_warp_gpu's flow-match branch in one call (build_wof_clips.py:70-73): first-pass flow, corners of the
 * first frame (1000 / 0.001 / 3, selection on host threads), the corners moved by the flow, RANSAC homography (threshold 1 px,
 * `hypotheses` samples from `seed`, refit on the inliers; identity unless > 50 matches and > 25 inliers), second frame warped back,
 * flow again.  The frames go up once and the first-pass fields stay on the device.  Outputs as vq_flow_tvl1 (any may be NULL), plus
 * the homographies frames0 -> frames1 [n][9], the match and inlier counts [n]. */
int vq_flow_warped(vq_flow* flow, const uint8_t* frames0, const uint8_t* frames1, int32_t n_pairs, uint32_t seed, int32_t hypotheses,
                   float* u1_host, float* u2_host, uint8_t* flow_x_host, uint8_t* flow_y_host, double* h_host, int32_t* matches_host,
                   int32_t* inliers_host, void* hip_stream);

/* Camera-motion estimation, the flow-match branch of extract_warp_gpu's "warp" step (improved dense trajectories, Wang &
 * Schmid 2013, as dense_flow applies it): Shi-Tomasi corners of the first frame (cv::goodFeaturesToTrack semantics: 3x3 block,
 * Sobel 3, candidates = 3x3 local maxima above quality * strongest, strongest first, none closer than min_distance to a kept
 * one; dense_flow asks for 1000 / 0.001 / 3), each matched to corner + first-pass flow at the corner, then a RANSAC
 * homography over the matches.  The second frame warped by the INVERSE of that homography (vq_flow_tvl1's homographies
 * argument) and the flow computed again is the "warped" flow.  The SURF matches the binary merges in are not built.
 * frames: n frames of the handle's h x w; corners_host [n][max_corners][2] (x, y), counts_host [n]. */
int vq_flow_good_features(vq_flow* flow, const uint8_t* frames, int32_t frames_on_device, int32_t n, int32_t max_corners, float quality,
                          float min_distance, float* corners_host, int32_t* counts_host, void* hip_stream);
/* n independent match sets src -> dst ([n][max_points][2] floats, counts [n]).  `hypotheses` 4-point samples per set, drawn
 * with a counter hash of (seed, set, hypothesis) so the result does not depend on scheduling; a sample whose two
 * quadrilaterals are oriented differently is skipped; score = matches with forward reprojection error <= threshold px;
 * winner = most inliers, first among equals.  refit != 0: the returned matrix is the normalised least-squares fit to the
 * winner's inliers (cv::findHomography re-estimates likewise; its final Levenberg-Marquardt polish is not applied).
 * h_host [n][9] row-major (identity when a set has fewer than 4 matches or no valid sample), inliers_host [n],
 * winner_host [n] (hypothesis index, -1 = none; optional), mask_host [n][max_points] (optional). */
int vq_flow_ransac_homography(vq_flow* flow, const float* src_host, const float* dst_host, const int32_t* counts_host, int32_t n,
                              int32_t max_points, float threshold, int32_t hypotheses, uint32_t seed, int32_t refit, double* h_host,
                              int32_t* inliers_host, int32_t* winner_host, uint8_t* mask_host, void* hip_stream);

/* ------------------------------------------------------------------------------------------
 * Comm group: one process per GPU, RCCL over xGMI
 * ------------------------------------------------------------------------------------------
 * The reference's only exchange is multiprocessing.Pool pickling per-clip features back to the parent
 * (src/features_GPU_compute/calcSig_wOF.py:204-210); these entry points are what a non-torch host (cgo, JNI, plain C)
 * binds to drive the sharded path.  librccl is resolved at run time (VQ_RCCL_LIB, else the RCCL already mapped into
 * the process, else librccl.so.1); VQ_E_UNSUPPORTED if there is none.  A torch host can keep using torch.distributed
 * (backend "nccl" = the same RCCL): the Python package does. */
typedef struct vq_comm vq_comm;
#define VQ_COMM_ID_BYTES 128
/* Rank 0 creates the 128-byte rendezvous id and hands it to the other ranks by any host-side means. */
int vq_comm_unique_id(void* id_out);
/* Collective over all ranks.  Makes `device` current on the calling thread and binds the communicator to it. */
int vq_comm_init(int32_t rank, int32_t world, const void* rccl_unique_id, int32_t device, vq_comm** out);
int vq_comm_destroy(vq_comm* comm);
int vq_comm_info(vq_comm* comm, int32_t* rank, int32_t* world, int32_t* device);
/* Half A -> half B hand-off: every rank contributes block_bytes of device memory (its [clips of the rank][S][1024]
 * feature block, padded to the largest shard); all_dev receives [world][block_bytes] in rank order = global clip order
 * for contiguous shards.  One all-gather on hip_stream, asynchronous. */
int vq_allgather_features(vq_comm* comm, const void* block_dev, int64_t block_bytes, void* all_dev, void* hip_stream);
/* Sharded scan: gathers the score slice of every rank's row shard (db's scores[n], zero-padded to slice_rows) into
 * all_scores_dev [world][slice_rows] fp64 on every rank -- N x 8 bytes cross xGMI, never the features.
 * Stream contract: vq_db_scan only ENQUEUES on the database handle's stream (vq_db_set_stream); this call orders its copy
 * of the scores behind that scan with an event, whatever hip_stream is, and the all-gather runs on hip_stream.  VQ_E_STATE
 * when the handle holds no scores (no scan / rescore since the last query).  Takes the handle's lock for the copy. */
int vq_allgather_scores(vq_comm* comm, vq_db* db, int64_t slice_rows, double* all_scores_dev, void* hip_stream);
/* The 80 KB query block t[S][E][D] (or any small device buffer) from `root` to every rank, in place. */
int vq_broadcast_query(vq_comm* comm, void* buf_dev, int64_t bytes, int32_t root, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* VQ_AMD_H */
