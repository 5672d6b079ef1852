// lfdmi.hip -- host side of liblfdmi.so: context/workspace, launch sequencing, C-ABI.
// Device code lives in k_image.h (dense image stages), k_ccl.h (hysteresis + contour
// topology), k_rect.h (minAreaRect / fillPoly), k_hough.h (HoughLines + check_theta).
// Written for gfx950 only; build with -ffp-contract=off (float32 evaluation order is part of
// the result: Hough vote bins, rotating calipers).
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <pthread.h>
#include <sched.h>
#include <string.h>

#include <atomic>
#include <condition_variable>
#include <string>
#include <thread>
#include <mutex>
#include <memory>
#include <vector>

#include "../../include/lfdmi.h"
#include "common.h"
#include "k_ccl.h"
#include "k_hough.h"
#include "k_image.h"
#include "k_rect.h"
#include "k_frame.h"
#include "fits_reader.h"

#define LFD_PI 3.1415926535897932384626433832795
#define MAX_ANGLES 4096 // rows of the cos/sin table (theta >= pi/4096)

// one timing slot per kernel (HIP events on the launch stream, see lfdmi_enable_timing)
enum {
    KID_REMOVESTARS = 0, KID_PREP_HIST, KID_LUT, KID_ERODE, KID_DILATE, KID_CANNY_NMS, KID_RUNS_INIT_FG,
    KID_RUNS_MERGE8, KID_RUNS_FLATTEN_FG, KID_EDGE, KID_RUNS_INIT_BG, KID_RUNS_MERGE4, KID_RUNS_FLATTEN_BG,
    KID_KEYS, KID_EXTREMES, KID_RECTS, KID_FILL, KID_PIXLIST, KID_VOTE, KID_PEAKS, KID_TOPK, KID_SORT,
    KID_FINALIZE, KID_DILATE_CANNY, KID_FRAME_FG, KID_FRAME_BG, KID_FRAME_KEYS, KID_PREP_ERODE, KID_PREP_DUAL, KID_BITS_ERODE, KID_MISC, TG_COUNT
};
static const char *const KID_NAMES[TG_COUNT] = {
    "k_removestars", "k_prep_hist", "k_lut", "k_morph(erode)", "k_morph(dilate)", "k_canny_nms", "k_runs_init(fg)",
    "k_runs_merge8", "k_runs_flatten(fg)", "k_edge_from_cand", "k_runs_init(bg)", "k_runs_merge4_bg",
    "k_runs_flatten(bg)", "k_keys", "k_extremes", "k_rects", "k_fill_quads", "k_pixlist", "k_hough_vote",
    "k_hough_peaks", "k_hough_topk", "k_hough_sort", "k_finalize", "k_dilate_canny", "k_frame_fg", "k_frame_bg",
    "k_frame_keys", "k_prep_erode", "k_prep_dual", "k_bits_erode", "misc"};

struct TimedSpan { hipEvent_t a, b; int group, pass, det; };

struct lfdmi_ctx {
    int device = 0, H = 0, W = 0, G = 0, wq = 0;
    size_t N = 0;
    hipStream_t stream = nullptr;
    bool own_stream = true;
    hipStream_t side[2] = {nullptr, nullptr}; // the tall-key rectangle kernels run beside the short-key one
    hipEvent_t ev_fork = nullptr, ev_join[2] = {nullptr, nullptr};
    std::string err;
    // dense images
    uint8_t *gray = nullptr, *tmp = nullptr, *equ = nullptr, *lut = nullptr, *mask = nullptr;
    int *hist = nullptr;
    u64 *candb = nullptr, *strongb = nullptr, *edgeb = nullptr, *equb = nullptr, *boxb = nullptr;
    // run labels (indexed by pixel index of a run start)
    // run tables (compact run ids, run_cap entries per slot) and per-word run-count scans
    int *Lf = nullptr, *YMf = nullptr, *FLf = nullptr, *Lb = nullptr, *YMb = nullptr, *FLb = nullptr;
    int *SBf = nullptr, *SBb = nullptr, *PAb = nullptr, *ROWf = nullptr, *ROWb = nullptr;
    int *scanf_ = nullptr, *scanb_ = nullptr;
    int run_cap = 0;
    int4 *keys = nullptr;
    int *bigkeys = nullptr, *medkeys = nullptr;
    int *wl_fg = nullptr, *wl_bg = nullptr; // work lists of active bit-row words
    int2 *rowext = nullptr;
    int *quads = nullptr;
    uint32_t *pix_equ = nullptr, *pix_box = nullptr;
    int *accum = nullptr;
    u64 *peaks = nullptr;
    float *lines = nullptr, *tab = nullptr;
    int *counters = nullptr, *need_dim = nullptr;
    uint8_t *zero_block = nullptr;     // cellbm | hist | counters
    size_t zero_bytes = 0, zero_counters_off = 0;
    // second set for the dual front end (lfdmi_detect_batch): the dim pass's histogram and cell bitmap are produced during the
    // bright pass's front end and must survive the zeroing at the start of the dim pass
    uint8_t *zero_block2 = nullptr;    // cellbm2 | hist2
    size_t zero_bytes2 = 0;
    u64 *cellbm2 = nullptr;            // cell occupancy of the eroded image (dual front end; k_morph_rect_v's output marks)
    u64 *candmask = nullptr;           // per 64 x 32 tile: the wide erosion can leave something there (k_erode_cand), 2 u64 per tile row
    u64 *fullbits = nullptr;           // one bit per aligned 4-pixel word of the 8-bit image: all four non-zero (k_prep_hist -> wide erosions)
    int *hist2 = nullptr;
    bool fuse_dual = false;            // LFDMI_FUSE_DUAL=1: one sweep over the float frames feeds both passes (measured slower:
                                       // the band kernel is bound by its instruction stream, not by HBM; kept for experiments)
    u64 *dbits = nullptr, *nzd = nullptr; // one bit per pixel each: dim value = bright value + bit; dim value non-zero (k_prep_hist<1, true> -> k_bits_erode)
    bool sparse_erode_fill = true;     // LFDMI_SPARSE_ERODE_FILL=0: a wide erosion zero-fills its whole output plane
    bool delta_dim = true;             // LFDMI_DELTA_DIM=0: the dim pass of lfdmi_detect_batch converts the float frames again
    int delta_state = 0;               // 1: this bright pass also writes dbits / hist2; 2: this dim pass starts from them
    int dual_state = 0;                // 0: none; 1: the next run_front is a bright pass that also feeds the dim pass; 2: dim pass already fed
    int2 *rsa = nullptr;               // k_frame_contours: (row slot, component) per candidate run, FRAME_RUNCAP per slot
    long long *prof = nullptr;         // LFDMI_FRAME_PROFILE=1: per-frame phase clocks of k_frame_contours (developer tool)
    u64 *cellbm = nullptr;             // cell occupancy of the last prep output, bm_bands x CELLBM_WORDS words per slot
    int bm_bands = 0;
    bool use_cellbm = true;
    int4 *segcnt = nullptr;            // per 64-word segment: run starts, fg / bg list entries (then their exclusive sums)
    int *fg_keys = nullptr;            // per slot: k_frame_fg made the outer-border keys and their extremes (k_frame.h)
    bool fg_keys_on = true;            // LFDMI_FG_KEYS=0: k_frame_contours makes every key itself, as before round 4
    int *fb_fg = nullptr, *fb_bg = nullptr; // per slot: frame left to the multi-workgroup run kernels (k_frame.h)
    int *perm = nullptr;               // k_active_perm: the current pass's active slots first (XCD balance of the per-frame launches)
    const int *perm_cur = nullptr;     // perm while a pass with an `active` mask runs, nullptr otherwise
    bool use_perm = true;              // LFDMI_PERM=0: frame slot == workgroup index as before
    int *perm_tiles = nullptr;         // k_tile_perm: frames by active tiles, dealt to the XCDs in snake order (the tile kernel's frame list)
    bool use_tile_perm = true;         // LFDMI_TILE_PERM=0
    bool sky_fast = true;              // LFDMI_SKY_FAST=0: the bright sweep without its all-sky shortcut
    bool vote_classes = true;          // LFDMI_VOTE_CLASSES=0: one chunk list per image (no longer cut for the mid-angle slabs)
    bool rs_fold_on = true;            // LFDMI_RS_FOLD=0: lfdmi_detect_batch zero-fills remove_stars' squares before the sweep instead of masking them in it
    bool rs_fold = false;              // (this chunk: the sweep masks the squares of rs_boxes; rs_count_dev / rs_max_obj describe them)
    const int *rs_count_dev = nullptr; int rs_max_obj = 0;
    int rs_fill_at = 13;               // LFDMI_RS_FILL_AT: where the deferred zero fill of device-resident frames is enqueued: 10 x pass + stage; 13 = before the DIM pass's k_frame_fg (3: the bright pass's)
                                       // (round 4: k_frame_contours alone, point 4, has become shorter than the fill, whose tail then hit the rectangle kernels)
    int rs_fill_at0 = 3, rs_fill_part = 0;  // LFDMI_RS_FILL_AT0: an earlier point for the first half of the objects (3: the bright pass's k_frame_fg; -1: one launch)
    float *rs_fill_frames = nullptr;   // (pending deferred fill: frames, nc, h, w)
    int rs_fill_nc = 0, rs_fill_h = 0, rs_fill_w = 0;
    hipEvent_t ev_rsfill = nullptr;
    bool rs_fill_inflight = false;
    bool scan_fused = true;            // LFDMI_SCAN_FUSED=0: the run scans as three launches (count, bases, write)
    u64 *scan_partial = nullptr;       // G x SCAN_MAX_BLK: k_scan_fused's per-workgroup totals + epoch marks
    int scan_epoch = 0;
    int scan_spin = 4096;              // polls a workgroup of k_scan_fused waits for a predecessor's totals (~0.5 ms; LFDMI_SCAN_SPIN, tests: 0)
    bool rects_prep = true;            // LFDMI_RECTS_PREP=0: the wave-per-key rectangle kernels scan their hulls sequentially
    bool vote_balance = true;          // LFDMI_VOTE_BALANCE=0: a fixed number of list pieces per image in the vote kernel
    lfdmi_result *res_dev = nullptr;   // G x LFDMI_MAX_SCALES records (one block of G per Hough scale)
    void *res_host = nullptr;          // page-locked: G flags + G records of the chunk just finished (lfdmi_detect_batch)
    // Workspace sizing (include/lfdmi.h: lfdmi_caps).  A compact context keeps a worst-case one for single frames
    // (`spill`, created on first use): a frame whose tables overflow here (per-frame LFDMI_ERR_CAPACITY) is run
    // again there, so no input fails for lack of table space.
    bool worst = false;                // every table at its theoretical maximum (the spill workspace itself)
    lfdmi_ctx *spill = nullptr;
    long long n_spilled = 0;           // frames re-run through the spill workspace since creation
    // lfdmi_get_stats: the other things that cost time without changing a result
    long long n_scan_giveups = 0, n_general_reruns = 0, n_general_chunks = 0, n_chunks = 0, n_cap_growths = 0;
    bool grow_on = true;               // LFDMI_GROW=0: overflowing frames always take the worst-case workspace, the tables never grow
    bool scan_fused_cfg = true;        // the one-launch scan is wanted (scan_fused: ... and in use right now)
    int scan_quiet = 0, scan_rearm = 64; // chunks since the look-back last gave up; chunks after which it is tried again (doubles per give-up)
    size_t bytes = 0;                  // device bytes of the workspace (dmalloc)
    double min_rho = 1.0;              // accumulators / peak lists sized for HoughLines rho >= min_rho
    void *scratch = nullptr;           // stand-alone HoughLines: sorted lines / untransposed accumulator
    size_t scratch_bytes = 0;
    int last_h = 0, last_w = 0;        // shape of the last call (lfdmi_get_stage)
    int quiet_chunks = 0;              // chunks since a frame last needed the general run kernels (general_seen decays)
    // host-frame feed of lfdmi_detect_batch: two pinned staging buffers filled by host threads, two device buffers, a copy
    // stream; chunk k+1 crosses PCIe while chunk k is being processed
    void *feed_pin[2] = {nullptr, nullptr}, *feed_dev[2] = {nullptr, nullptr};
    size_t feed_bytes = 0, feed_pin_bytes = 0; // capacity of each of the two device / pinned host buffers
    bool feed_cpus_known = false;
    hipStream_t feed_copy = nullptr, feed_copy2 = nullptr; // (LFDMI_FEED_STREAMS=2: pieces alternate between two copy streams)
    hipEvent_t feed_mid = nullptr;
    int feed_streams = 2;
    hipEvent_t feed_up[2] = {nullptr, nullptr};
    int feed_threads = 4;              // host threads copying a chunk into the pinned buffer (LFDMI_FEED_THREADS)
    size_t feed_chunk_bytes = 800u << 20; // largest feed chunk (LFDMI_FEED_MB): 64 SDSS frames, 11 frames of 4096 x 4096
    std::vector<int> feed_cpus;        // CPUs local to the GPU (numa_cpus): feed / blot threads and the pinned buffers are bound to them
    int fail_chunk = -1;               // lfdmi_debug_fail_chunk: the next lfdmi_detect_batch call fails at the top of this chunk (tests)
    int4 *rs_sboxes = nullptr;         // crowded catalogues: the squares sorted by first row + rowstart (k_rs_sort); rs_sorted: this chunk uses them
    int *rs_rowstart = nullptr;
    size_t rs_rowstart_cap = 0;
    bool rs_sorted = false;
    int rs_sort_min = RS_SORT_MIN;     // LFDMI_RS_SORT_MIN: catalogues with more objects per frame take the sorted path (tests: 0)
    int rs_hmax = 0;
    int4 *rs_boxes = nullptr;          // remove_stars squares of the chunk (k_rs_boxes -> k_rs_fill, and the host's own blotting)
    size_t rs_boxes_cap = 0;
    void *stage = nullptr;
    size_t stage_bytes = 0;
    void *cat_dev = nullptr;
    size_t cat_bytes = 0;
    int key_cap = 0, slot_cap = 0;
    size_t acc_cap = 0, peak_cap = 0, list_cap = 0, peak_worst = 0;
    // cached Hough tables: the last few (rho, theta, shape) combinations keep their device trig table, so a
    // multi-scale pass (rho = 20, 10, 5 on the same image) does not rebuild and re-upload tables per launch
    struct HoughTab {
        double rho = -1, theta = -1;
        int h = 0, w = 0, numangle = 0, numrho = 0;
        long long stamp = 0;
        std::vector<float> host;       // host copy (slab ranges of the vote kernel; source of the upload)
    } tabs[LFDMI_MAX_SCALES];
    long long tab_clock = 0;
    int tab_cur = 0;                   // entry the next Hough launch uses: device table = tab + tab_cur * 2 * MAX_ANGLES
    int numangle = 0, numrho = 0;
    // timing
    bool timing = false;
    uint64_t timing_mask = 0;          // non-zero: only these timing slots' launches are bracketed (lfdmi_timing_select)
    std::vector<TimedSpan> spans;
    std::vector<hipEvent_t> ev_pool;
    float t_ms[TG_COUNT] = {0};
    int t_n[TG_COUNT] = {0};
    long long t_units[TG_COUNT] = {0}; // frames (images) the timed launches actually worked on
    int stage_mode = -1;               // lfdmi_set_stage_images: -1 the per-pass entry points keep the 8-bit stage images, lfdmi_detect_batch
                                       // does not; 0 never; 1 always
    bool stages_valid = false;         // the last call kept them (lfdmi_get_stage)
    bool eroded_valid = false;         // the last dim front end wrote its whole eroded plane (LFDMI_STAGE_ERODED readable even without
                                       // the other stage images: the parity tests of lfdmi_detect_batch's bit-plane front end)
    int vote_split = 4;                // pieces a frame's Hough list is cut into at most
    int pe_rows = 12;                  // rows per band of k_prep_erode
    bool fuse_prep_erode = true;       // dim pass: prep + histogram + erosion in one kernel (LFDMI_FUSE_PREP_ERODE=0: separate)
    // general run kernels beside the per-frame ones: launched once this context has met a frame that needs them
    // (or always, for the per-operator entry points and with LFDMI_FRAME_CCL=0)
    bool general_seen = false, general_on = true;
    bool frame_ccl = true;             // per-frame LDS connectivity kernels (k_frame.h)
    int frame_runcap = FRAME_RUNCAP;   // runs per frame they take (LFDMI_FRAME_RUNCAP lowers it: tests of the fallback path)
    int frame_dbg = 0;                 // LFDMI_FRAME_DBG: developer attribution switches of k_frame_contours (wrong results)
    int frame_lds = FRAME_RUNCAP;      // entries of their LDS label table (LFDMI_FRAME_LDS: a smaller table leaves LDS to other kernels
                                       // on the CU; frames with more runs take the general kernels)
    int *tile_list = nullptr;          // per slot: active 64 x 16 tiles of the pass image (k_dc_tiles -> k_dilate_canny_t)
    int tile_cap = 0;
    bool dc_specialize = true;         // LFDMI_DC_SPECIALIZE=0: the run-time-size instantiation of k_dilate_canny_t for every kernel size
    bool dc_profile = false;           // LFDMI_DC_PROFILE=1: stage clocks of k_dilate_canny_t into `prof` (developer tool)
    bool dc_tilelist = true;           // LFDMI_DC_TILELIST=0: the strip-walking kernel k_dilate_canny_w instead
    int dc_parts = 0;                  // waves per frame of k_dilate_canny_t (0: tiles per frame / 8, at most 256; LFDMI_DC_PARTS)
    int dc_substrips = 4;              // strips a wave of k_dilate_canny_w walks one after the other
    int dc_strip = 8;                  // tiles per wave strip in k_dilate_canny_w
    bool keep_equ = true;              // write the equalised+dilated stage image (off in lfdmi_detect_batch)
    int cur_pass = 0;                  // 0 = bright / stand-alone operator, 1 = dim pass of detect_batch
    int *pass_flags = nullptr;         // per slot: bit0 bright detection, bit1 dim pass ran, bit2 dim detection
    std::vector<void *> allocs;
};

static int fail(lfdmi_ctx *c, int code, const std::string &msg) {
    if (c) c->err = msg;
    return code;
}

#define HIPCHK(expr)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(ctx, LFDMI_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
#define KCHK(name)                                                                             \
    do {                                                                                       \
        hipError_t e_ = hipGetLastError();                                                     \
        if (e_ != hipSuccess)                                                                  \
            return fail(ctx, LFDMI_ERR_HIP, std::string("launch ") + name + ": " + hipGetErrorString(e_)); \
    } while (0)
#define RET(expr)                   \
    do {                            \
        int rc_ = (expr);           \
        if (rc_) return rc_;        \
    } while (0)

struct Span {
    lfdmi_ctx *c;
    int idx = -1;
    hipStream_t st;
    Span(lfdmi_ctx *ctx, int group, int det = 0, hipStream_t on = nullptr) : c(ctx), st(on ? on : ctx->stream) {
        if (!c->timing || (c->timing_mask && !((c->timing_mask >> group) & 1ull))) return;
        TimedSpan s;
        s.group = group;
        s.pass = ctx->cur_pass;
        s.det = det;
        for (hipEvent_t *e : {&s.a, &s.b}) {
            if (!c->ev_pool.empty()) { *e = c->ev_pool.back(); c->ev_pool.pop_back(); }
            else hipEventCreate(e);
        }
        hipEventRecord(s.a, st);
        c->spans.push_back(s);
        idx = (int)c->spans.size() - 1;
    }
    ~Span() {
        if (idx >= 0) hipEventRecord(c->spans[idx].b, st);
    }
};

// n_act[p] / n_det[p]: frames of the chunk that pass p worked on / that reached its Hough stage
static void collect_spans(lfdmi_ctx *c, const int n_act[2], const int n_det[2]) {
    for (auto &s : c->spans) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
            c->t_ms[s.group] += ms;
            c->t_n[s.group]++;
            c->t_units[s.group] += s.det ? n_det[s.pass] : n_act[s.pass];
        }
        c->ev_pool.push_back(s.a);
        c->ev_pool.push_back(s.b);
    }
    c->spans.clear();
}

template <typename T> static int dmalloc(lfdmi_ctx *ctx, T **p, size_t count) {
    void *q = nullptr;
    HIPCHK(hipMalloc(&q, count * sizeof(T)));
    ctx->allocs.push_back(q);
    ctx->bytes += count * sizeof(T);
    *p = (T *)q;
    return 0;
}

static size_t next_pow2(size_t v) { size_t p = 1; while (p < v) p <<= 1; return p; }

extern "C" int lfdmi_version(void) { return LFDMI_VERSION; }

extern "C" void lfdmi_hough_dims(int h, int w, double rho_d, double theta_d, int *numangle, int *numrho) {
    float rho = (float)rho_d, theta = (float)theta_d;
    *numangle = (int)lrint((LFD_PI - 0.0) / theta);
    *numrho = (int)lrint(((w + h) * 2 + 1) / rho);
}

// theoretical maxima of the per-frame tables (a checkerboard image)
static void worst_caps(int h, int w, lfdmi_caps *c) {
    size_t N = (size_t)h * w;
    c->run_cap = (int)((size_t)h * (w / 2 + 1) + 16);
    c->key_cap = (int)(N / 2 + 16);
    c->slot_cap = (int)(2 * N + 4 * (size_t)h + 16);
    c->list_cap = (int)N;
    c->peak_cap = 0; // numangle * numrho at min_rho
    c->min_rho = 1.0;
}

extern "C" void lfdmi_default_caps(int max_h, int max_w, lfdmi_caps *out) {
    // tools/cap_survey.py (SDSS batch, LSST-size dim / bright passes): runs <= N/165, contours <= N/3100, contour
    // rows <= N/228, Hough chunks <= N/282, peaks <= 9 000 per image.  Defaults leave ~10x headroom; the floors keep
    // small test images (dense random noise) out of the spill path.
    size_t N = (size_t)max_h * max_w;
    lfdmi_caps wc;
    worst_caps(max_h, max_w, &wc);
    auto pick = [](size_t want, size_t floor_, int worst) { size_t v = want > floor_ ? want : floor_; return (int)(v < (size_t)worst ? v : (size_t)worst); };
    out->run_cap = pick(N / 16, 16384, wc.run_cap);
    out->key_cap = pick(N / 256, 4096, wc.key_cap);
    out->slot_cap = pick(N / 16, 16384, wc.slot_cap);
    out->list_cap = pick(N / 16, 16384, wc.list_cap);
    out->peak_cap = 65536;
    out->min_rho = 5.0;
}

static int create_impl(int device, int max_h, int max_w, int max_inflight, const lfdmi_caps *caps_in, lfdmi_ctx **out) {
    if (!out || max_h <= 0 || max_w <= 0 || max_inflight <= 0 || max_h > 8191 || max_w > 8191) return LFDMI_ERR_ARG;
    lfdmi_ctx *ctx = new lfdmi_ctx();
    *out = ctx;
    ctx->device = device; ctx->H = max_h; ctx->W = max_w; ctx->G = max_inflight;
    if (const char *e = getenv("LFDMI_FRAME_CCL")) ctx->frame_ccl = atoi(e) != 0;
    if (const char *e = getenv("LFDMI_FUSE_PREP_ERODE")) ctx->fuse_prep_erode = atoi(e) != 0;
    if (const char *e = getenv("LFDMI_FUSE_DUAL")) ctx->fuse_dual = atoi(e) != 0;
    if (const char *e = getenv("LFDMI_DELTA_DIM")) ctx->delta_dim = atoi(e) != 0;
    if (const char *e = getenv("LFDMI_SPARSE_ERODE_FILL")) ctx->sparse_erode_fill = atoi(e) != 0;
    if (const char *e = getenv("LFDMI_VOTE_SPLIT")) { int v = atoi(e); if (v >= 1 && v <= 16) ctx->vote_split = v; }
    if (const char *e = getenv("LFDMI_PE_ROWS")) { int v = atoi(e); if (v >= 2 && v <= 64) ctx->pe_rows = v; }
    if (const char *e = getenv("LFDMI_FRAME_DBG")) ctx->frame_dbg = atoi(e);
    if (const char *e = getenv("LFDMI_FRAME_RUNCAP")) { int v = atoi(e); if (v >= 0 && v < FRAME_RUNCAP) ctx->frame_runcap = v; }
    if (const char *e = getenv("LFDMI_FRAME_LDS")) {
        int v = atoi(e);
        if (v >= 32 && v <= FRAME_RUNCAP) { ctx->frame_lds = (v + 31) & ~31; ctx->frame_runcap = std::min(ctx->frame_runcap, ctx->frame_lds); }
    }
    if (const char *e = getenv("LFDMI_DC_SUBSTRIPS")) { int v = atoi(e); if (v >= 1 && v <= 64) ctx->dc_substrips = v; }
    if (const char *e = getenv("LFDMI_DC_STRIP")) { int v = atoi(e); if (v >= 1 && v <= DCW_MAXS) ctx->dc_strip = v; } // tuning knob
    if (const char *e = getenv("LFDMI_DC_TILELIST")) ctx->dc_tilelist = atoi(e) != 0;
    if (const char *e = getenv("LFDMI_DC_PARTS")) { int v = atoi(e); if (v >= 1 && v <= 4096) ctx->dc_parts = v; }
    if (const char *e = getenv("LFDMI_FEED_STREAMS")) { int v = atoi(e); if (v == 1 || v == 2) ctx->feed_streams = v; }
    if (const char *e = getenv("LFDMI_FEED_THREADS")) { int v = atoi(e); if (v >= 1 && v <= 64) ctx->feed_threads = v; }
    if (const char *e = getenv("LFDMI_FEED_MB")) { int v = atoi(e); if (v >= 0 && v <= 8192) ctx->feed_chunk_bytes = (size_t)v << 20; } // 0: plain staging
    ctx->N = (size_t)max_h * max_w;
    ctx->wq = LFD_WQ(max_w);
    size_t N = ctx->N, G = (size_t)max_inflight, BW = (size_t)max_h * ctx->wq;
    // capacities: the caller's, the defaults, or (fields <= 0, LFDMI_WORST_CASE=1) the theoretical maxima
    lfdmi_caps wc, caps;
    worst_caps(max_h, max_w, &wc);
    if (caps_in) caps = *caps_in; else lfdmi_default_caps(max_h, max_w, &caps);
    if (const char *e = getenv("LFDMI_WORST_CASE")) if (atoi(e)) caps = wc;
    auto cap = [](int v, int worst) { return (v <= 0 || v > worst) ? worst : v; };
    ctx->run_cap = cap(caps.run_cap, wc.run_cap);
    ctx->key_cap = cap(caps.key_cap, wc.key_cap);
    ctx->slot_cap = cap(caps.slot_cap, wc.slot_cap);
    ctx->list_cap = (size_t)cap(caps.list_cap, wc.list_cap);
    ctx->min_rho = caps.min_rho >= 1.0 ? caps.min_rho : 1.0;
    int na, nr;
    lfdmi_hough_dims(max_h, max_w, ctx->min_rho, LFD_PI / 180, &na, &nr);
    ctx->acc_cap = (size_t)(na + 2) * (nr + 2);
    size_t peak_worst = next_pow2((size_t)na * nr);
    ctx->peak_cap = caps.peak_cap > 0 ? next_pow2((size_t)caps.peak_cap) : peak_worst;
    if (ctx->peak_cap > peak_worst) ctx->peak_cap = peak_worst;
    ctx->peak_worst = peak_worst;
    ctx->worst = ctx->run_cap == wc.run_cap && ctx->key_cap == wc.key_cap && ctx->slot_cap == wc.slot_cap &&
                 ctx->list_cap == (size_t)wc.list_cap && ctx->peak_cap == peak_worst && ctx->min_rho <= 1.0;
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    for (int i = 0; i < 2; i++) {
        HIPCHK(hipStreamCreateWithFlags(&ctx->side[i], hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&ctx->ev_join[i], hipEventDisableTiming));
    }
    HIPCHK(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&ctx->ev_rsfill, hipEventDisableTiming));
    RET(dmalloc(ctx, &ctx->gray, G * N));
    RET(dmalloc(ctx, &ctx->tmp, G * N));
    RET(dmalloc(ctx, &ctx->equ, G * N));
    RET(dmalloc(ctx, &ctx->lut, G * 256));
    RET(dmalloc(ctx, &ctx->mask, (size_t)LFDMI_MAX_MORPH_K * LFDMI_MAX_MORPH_K * 2));
    RET(dmalloc(ctx, &ctx->candb, G * BW));
    RET(dmalloc(ctx, &ctx->strongb, G * BW));
    RET(dmalloc(ctx, &ctx->edgeb, G * BW));
    RET(dmalloc(ctx, &ctx->equb, G * BW));
    RET(dmalloc(ctx, &ctx->boxb, G * BW));
    RET(dmalloc(ctx, &ctx->dbits, G * BW));
    RET(dmalloc(ctx, &ctx->nzd, G * BW));
    for (int **p : {&ctx->Lf, &ctx->YMf, &ctx->FLf, &ctx->Lb, &ctx->YMb, &ctx->FLb, &ctx->SBf, &ctx->SBb, &ctx->PAb,
                    &ctx->ROWf, &ctx->ROWb})
        RET(dmalloc(ctx, p, G * ctx->run_cap));
    if (const char *e = getenv("LFDMI_DC_PROFILE")) ctx->dc_profile = atoi(e) != 0;
    if (const char *e = getenv("LFDMI_DC_SPECIALIZE")) ctx->dc_specialize = atoi(e) != 0;
    if (getenv("LFDMI_FRAME_PROFILE") || ctx->dc_profile) RET(dmalloc(ctx, &ctx->prof, G * 16));
    RET(dmalloc(ctx, &ctx->rsa, G * FRAME_RUNCAP));
    ctx->bm_bands = (max_h + CELLBM_ROWS - 1) / CELLBM_ROWS;
    if (const char *e = getenv("LFDMI_CELLBM")) ctx->use_cellbm = atoi(e) != 0;
    ctx->tile_cap = ((max_h + DCW_TH - 1) / DCW_TH) * ((max_w + CANNY_TW - 1) / CANNY_TW);
    RET(dmalloc(ctx, &ctx->tile_list, G * ctx->tile_cap));
    RET(dmalloc(ctx, &ctx->segcnt, G * SCAN_MAX_SEG));
    RET(dmalloc(ctx, &ctx->perm, G));
    RET(dmalloc(ctx, &ctx->perm_tiles, G));
    if (const char *e = getenv("LFDMI_TILE_PERM")) ctx->use_tile_perm = atoi(e) != 0;
    if (const char *e = getenv("LFDMI_SKY_FAST")) ctx->sky_fast = atoi(e) != 0;
    if (const char *e = getenv("LFDMI_VOTE_CLASSES")) ctx->vote_classes = atoi(e) != 0;
    if (const char *e = getenv("LFDMI_VOTE_BALANCE")) ctx->vote_balance = atoi(e) != 0;
    if (const char *e = getenv("LFDMI_RECTS_PREP")) ctx->rects_prep = atoi(e) != 0;
    if (const char *e = getenv("LFDMI_RS_FOLD")) ctx->rs_fold_on = atoi(e) != 0;
    if (const char *e = getenv("LFDMI_RS_SORT_MIN")) ctx->rs_sort_min = atoi(e);
    if (const char *e = getenv("LFDMI_SCAN_FUSED")) ctx->scan_fused = atoi(e) != 0;
    ctx->scan_fused_cfg = ctx->scan_fused;
    if (const char *e = getenv("LFDMI_GROW")) ctx->grow_on = atoi(e) != 0;
    if (const char *e = getenv("LFDMI_SCAN_REARM")) ctx->scan_rearm = std::max(1, atoi(e));
    if (const char *e = getenv("LFDMI_SCAN_SPIN")) ctx->scan_spin = std::max(0, atoi(e));
    if (const char *e = getenv("LFDMI_SCAN_EPOCH0")) ctx->scan_epoch = atoi(e) & ((1 << 22) - 1); // (tests: start next to the 22-bit wrap)
    if (const char *e = getenv("LFDMI_RS_FILL_AT")) ctx->rs_fill_at = atoi(e);
    if (const char *e = getenv("LFDMI_RS_FILL_AT0")) ctx->rs_fill_at0 = atoi(e);
    if (const char *e = getenv("LFDMI_PERM")) ctx->use_perm = atoi(e) != 0;
    RET(dmalloc(ctx, &ctx->fb_fg, G));
    RET(dmalloc(ctx, &ctx->fg_keys, G));
    HIPCHK(hipMemsetAsync(ctx->fg_keys, 0, G * sizeof(int), ctx->stream));
    if (const char *e = getenv("LFDMI_FG_KEYS")) ctx->fg_keys_on = atoi(e) != 0;
    RET(dmalloc(ctx, &ctx->fb_bg, G));
    RET(dmalloc(ctx, &ctx->scanf_, G * BW));
    RET(dmalloc(ctx, &ctx->scanb_, G * BW));
    RET(dmalloc(ctx, &ctx->keys, G * ctx->key_cap));
    RET(dmalloc(ctx, &ctx->bigkeys, G * ctx->key_cap));
    RET(dmalloc(ctx, &ctx->medkeys, G * ctx->key_cap));
    RET(dmalloc(ctx, &ctx->wl_fg, G * BW));
    RET(dmalloc(ctx, &ctx->wl_bg, G * BW));
    RET(dmalloc(ctx, &ctx->rowext, G * ctx->slot_cap));
    RET(dmalloc(ctx, &ctx->quads, G * ctx->key_cap * 8));
    RET(dmalloc(ctx, &ctx->pix_equ, G * 2 * ctx->list_cap)); // two chunk lists per image (k_pixlist: class A / class B)
    RET(dmalloc(ctx, &ctx->pix_box, G * 2 * ctx->list_cap));
    RET(dmalloc(ctx, &ctx->accum, G * 2 * ctx->acc_cap));
    RET(dmalloc(ctx, &ctx->peaks, G * 2 * ctx->peak_cap));
    RET(dmalloc(ctx, &ctx->lines, G * 2 * LFDMI_MAX_SET_LINES * 2));
    RET(dmalloc(ctx, &ctx->tab, (size_t)LFDMI_MAX_SCALES * 2 * MAX_ANGLES));
    {   // counters, histograms and the cell bitmap start every pass at zero: one block, one fill per pass
        size_t nb_cnt = G * C_COUNT * sizeof(int), nb_hist = G * 256 * sizeof(int), nb_bm = G * ctx->bm_bands * CELLBM_WORDS * sizeof(u64);
        uint8_t *blk = nullptr;
        RET(dmalloc(ctx, &blk, nb_bm + nb_hist + nb_cnt));
        ctx->cellbm = (u64 *)blk;
        ctx->hist = (int *)(blk + nb_bm);
        ctx->counters = (int *)(blk + nb_bm + nb_hist);
        ctx->zero_block = blk;
        ctx->zero_bytes = nb_bm + nb_hist + nb_cnt;
        ctx->zero_counters_off = nb_bm + nb_hist;
        uint8_t *blk2 = nullptr;
        RET(dmalloc(ctx, &blk2, nb_bm + nb_hist));
        ctx->cellbm2 = (u64 *)blk2;
        ctx->hist2 = (int *)(blk2 + nb_bm);
        ctx->zero_block2 = blk2;
        ctx->zero_bytes2 = nb_bm + nb_hist;
        RET(dmalloc(ctx, &ctx->fullbits, G * (size_t)max_h * ((max_w + 255) >> 8)));
        RET(dmalloc(ctx, &ctx->candmask, G * (size_t)((max_h + MORPH_TH - 1) / MORPH_TH) * 2));
    }
    RET(dmalloc(ctx, &ctx->need_dim, G));
    RET(dmalloc(ctx, &ctx->scan_partial, G * SCAN_MAX_BLK));
    HIPCHK(hipMemsetAsync(ctx->scan_partial, 0, G * SCAN_MAX_BLK * sizeof(u64), ctx->stream));
    { // flags and records side by side: lfdmi_detect_batch fetches both with one copy into page-locked memory (res_host)
        char *blk = nullptr;
        RET(dmalloc(ctx, &blk, G * sizeof(int) + G * LFDMI_MAX_SCALES * sizeof(lfdmi_result)));
        ctx->pass_flags = (int *)blk;
        ctx->res_dev = (lfdmi_result *)(blk + G * sizeof(int));
        HIPCHK(hipHostMalloc(&ctx->res_host, G * sizeof(int) + G * sizeof(lfdmi_result), hipHostMallocDefault));
    }
    HIPCHK(hipFuncSetAttribute((const void *)k_hough_vote<6>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
    HIPCHK(hipFuncSetAttribute((const void *)k_hough_vote<5>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
    HIPCHK(hipFuncSetAttribute((const void *)k_hough_vote<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
    HIPCHK(hipFuncSetAttribute((const void *)k_hough_vote<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
    HIPCHK(hipFuncSetAttribute((const void *)k_hough_vote<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
    HIPCHK(hipFuncSetAttribute((const void *)k_hough_vote<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
    HIPCHK(hipFuncSetAttribute((const void *)k_hough_vote<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
    HIPCHK(hipFuncSetAttribute((const void *)k_rects_big, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
    HIPCHK(hipFuncSetAttribute((const void *)k_dilate_canny_v, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    HIPCHK(hipFuncSetAttribute((const void *)k_frame_fg, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024)); // (the kernel has a few hundred bytes of static LDS)
    HIPCHK(hipFuncSetAttribute((const void *)k_prep_erode<false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
    HIPCHK(hipFuncSetAttribute((const void *)k_prep_erode<false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
    HIPCHK(hipFuncSetAttribute((const void *)k_prep_erode<false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
    HIPCHK(hipFuncSetAttribute((const void *)k_prep_erode<false, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
    HIPCHK(hipFuncSetAttribute((const void *)k_prep_erode<true, LFDMI_PREP_BRIGHT_THEN_DIM>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
    HIPCHK(hipFuncSetAttribute((const void *)k_frame_contours, hipFuncAttributeMaxDynamicSharedMemorySize,
                               (FRAME_RUNCAP + 4 * (FRAME_RUNCAP / 32)) * (int)sizeof(int)));
    HIPCHK(hipFuncSetAttribute((const void *)k_morph_rect<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    HIPCHK(hipFuncSetAttribute((const void *)k_morph_rect<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    return 0;
}

extern "C" int lfdmi_ctx_create(int device, int max_h, int max_w, int max_inflight, lfdmi_ctx **out) {
    return create_impl(device, max_h, max_w, max_inflight, nullptr, out);
}
extern "C" int lfdmi_ctx_create_sized(int device, int max_h, int max_w, int max_inflight, const lfdmi_caps *caps, lfdmi_ctx **out) {
    return create_impl(device, max_h, max_w, max_inflight, caps, out);
}
extern "C" int64_t lfdmi_ctx_bytes(lfdmi_ctx *ctx) { return ctx ? (int64_t)(ctx->bytes + (ctx->spill ? ctx->spill->bytes : 0)) : 0; }
extern "C" int64_t lfdmi_spill_count(lfdmi_ctx *ctx) { return ctx ? ctx->n_spilled : 0; }
extern "C" int lfdmi_get_stats(lfdmi_ctx *ctx, int64_t *out, int n) {
    if (!ctx || !out || n < 0 || n > LFDMI_STAT_COUNT) return LFDMI_ERR_ARG;
    const int64_t v[LFDMI_STAT_COUNT] = {ctx->n_spilled, ctx->n_scan_giveups, ctx->n_general_reruns, ctx->n_general_chunks, ctx->n_chunks,
                                         ctx->n_cap_growths, ctx->scan_fused ? 1 : 0};
    for (int i = 0; i < n; i++) out[i] = v[i];
    return 0;
}

// The look-back of k_scan_fused gave up on a frame of this chunk (the GPU is shared with other work, see k_ccl.h): the context
// goes back to three launches per scan and the chunk is run again -- cheaper than the worst-case rerun of every flagged frame.
// The one-launch scan is tried again after scan_rearm quiet chunks (doubling with every give-up: a GPU that stays shared
// settles on the three launches, one that was shared for a moment gets its fast scan back).
static bool scan_gave_up(lfdmi_ctx *ctx, const int *flags, int nc) {
    bool gave = false;
    for (int i = 0; i < nc; i++) gave = gave || (flags[i] & PASS_FLAG_SCAN_GAVEUP);
    if (!gave || !ctx->scan_fused) return false;
    ctx->scan_fused = false;
    ctx->n_scan_giveups++;
    ctx->scan_quiet = 0;
    ctx->scan_rearm = std::min(ctx->scan_rearm * 2, 1 << 16);
    return true;
}
static void chunk_done(lfdmi_ctx *ctx) {
    ctx->n_chunks++;
    if (!ctx->scan_fused && ctx->scan_fused_cfg && ++ctx->scan_quiet >= ctx->scan_rearm) { ctx->scan_fused = true; ctx->scan_quiet = 0; }
}

// ---- growing the per-frame tables ----------------------------------------------------------------------------------------
// The default capacities (lfdmi_default_caps) are ~10x what sky frames use; a crowded field, a noisy or saturated frame can
// still overflow them.  Until round 4 every such frame was run again ALONE in the worst-case workspace (a whole pipeline of
// launches for one frame: milliseconds instead of microseconds), every time.  Now the context enlarges the tables that were
// asked for more than they hold (k_finalize leaves the frame's demands in its record) and runs the chunk again; the worst-case
// workspace remains for what no growth can satisfy (memory, the theoretical maxima).  LFDMI_GROW=0 switches it off.
template <typename T> static int regrow(lfdmi_ctx *ctx, T **p, size_t old_count, size_t new_count) {
    void *q = nullptr;
    HIPCHK(hipMalloc(&q, new_count * sizeof(T)));
    for (auto &a : ctx->allocs) if (a == (void *)*p) a = q;
    (void)hipFree(*p);
    ctx->bytes += (new_count - old_count) * sizeof(T);
    *p = (T *)q;
    return 0;
}

// true: tables were enlarged, run the chunk again
static bool grow_caps(lfdmi_ctx *ctx, const lfdmi_result *rec, int nc, size_t rec_stride = 1, int n_scales = 1) {
    if (!ctx->grow_on || ctx->worst) return false;
    long long need_run = 0, need_key = 0, need_slot = 0, need_list = 0, need_peak = 0;
    bool any = false;
    for (int s = 0; s < n_scales; s++)
        for (int i = 0; i < nc; i++) {
            const lfdmi_result &r = rec[(size_t)s * rec_stride + i];
            if (r.status != LFDMI_ERR_CAPACITY) continue;
            any = true;
            need_run = std::max<long long>(need_run, r.x1); need_key = std::max<long long>(need_key, r.y1);
            need_slot = std::max<long long>(need_slot, r.x2); need_list = std::max<long long>(need_list, r.y2);
            need_peak = std::max<long long>(need_peak, r.n_lines_equ);
        }
    if (!any) return false;
    lfdmi_caps wc;
    worst_caps(ctx->H, ctx->W, &wc);
    const size_t peak_worst = ctx->peak_worst;
    auto up = [](long long cap, long long need, long long worst, bool at_cap_means_more) -> long long {
        if (need < cap || (need == cap && !at_cap_means_more)) return cap;
        return std::min<long long>(worst, std::max<long long>(2 * cap, need + need / 4 + 64));
    };
    long long run = up(ctx->run_cap, need_run, wc.run_cap, false), key = up(ctx->key_cap, need_key, wc.key_cap, true);
    long long slot = up(ctx->slot_cap, need_slot, wc.slot_cap, false), list = up((long long)ctx->list_cap, need_list, wc.list_cap, false);
    long long peak = up((long long)ctx->peak_cap, need_peak, (long long)peak_worst, false);
    if (peak > (long long)ctx->peak_cap) peak = (long long)std::min(peak_worst, next_pow2((size_t)peak));
    // a frame that ran out of runs never reached its keys, slots and lists: those tables follow in proportion
    if (run > ctx->run_cap) {
        const double f = (double)run / ctx->run_cap;
        key = std::max(key, std::min<long long>(wc.key_cap, (long long)(ctx->key_cap * f)));
        slot = std::max(slot, std::min<long long>(wc.slot_cap, (long long)(ctx->slot_cap * f)));
        list = std::max(list, std::min<long long>(wc.list_cap, (long long)(ctx->list_cap * f)));
    }
    if (run == ctx->run_cap && key == ctx->key_cap && slot == ctx->slot_cap && list == (long long)ctx->list_cap && peak == (long long)ctx->peak_cap)
        return false; // (an overflow no table explains -- e.g. the look-back scan's flag: the worst-case workspace takes the frame)
    const size_t G = (size_t)ctx->G;
    const long long extra = (long long)G * ((run - ctx->run_cap) * 11 * 4 + (key - ctx->key_cap) * (16 + 4 + 4 + 32) + (slot - ctx->slot_cap) * 8 +
                                            (list - (long long)ctx->list_cap) * 4 * 4 + (peak - (long long)ctx->peak_cap) * 2 * 8);
    size_t mem_free = 0, mem_total = 0;
    if (hipSetDevice(ctx->device) != hipSuccess || hipMemGetInfo(&mem_free, &mem_total) != hipSuccess) return false;
    if (extra + (2ll << 30) > (long long)mem_free) return false; // (no room: the worst-case workspace, one frame at a time)
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return false;
    (void)hipStreamSynchronize(ctx->side[0]); (void)hipStreamSynchronize(ctx->side[1]);
    int rc = 0;
    if (run > ctx->run_cap) {
        for (int **p : {&ctx->Lf, &ctx->YMf, &ctx->FLf, &ctx->Lb, &ctx->YMb, &ctx->FLb, &ctx->SBf, &ctx->SBb, &ctx->PAb, &ctx->ROWf, &ctx->ROWb})
            rc = rc ? rc : regrow(ctx, p, G * ctx->run_cap, G * (size_t)run);
        if (!rc) ctx->run_cap = (int)run;
    }
    if (!rc && key > ctx->key_cap) {
        rc = regrow(ctx, &ctx->keys, G * ctx->key_cap, G * (size_t)key);
        rc = rc ? rc : regrow(ctx, &ctx->bigkeys, G * ctx->key_cap, G * (size_t)key);
        rc = rc ? rc : regrow(ctx, &ctx->medkeys, G * ctx->key_cap, G * (size_t)key);
        rc = rc ? rc : regrow(ctx, &ctx->quads, G * ctx->key_cap * 8, G * (size_t)key * 8);
        if (!rc) ctx->key_cap = (int)key;
    }
    if (!rc && slot > ctx->slot_cap) { rc = regrow(ctx, &ctx->rowext, G * ctx->slot_cap, G * (size_t)slot); if (!rc) ctx->slot_cap = (int)slot; }
    if (!rc && list > (long long)ctx->list_cap) {
        rc = regrow(ctx, &ctx->pix_equ, G * 2 * ctx->list_cap, G * 2 * (size_t)list);
        rc = rc ? rc : regrow(ctx, &ctx->pix_box, G * 2 * ctx->list_cap, G * 2 * (size_t)list);
        if (!rc) ctx->list_cap = (size_t)list;
    }
    if (!rc && peak > (long long)ctx->peak_cap) { rc = regrow(ctx, &ctx->peaks, G * 2 * ctx->peak_cap, G * 2 * (size_t)peak); if (!rc) ctx->peak_cap = (size_t)peak; }
    if (rc) return false; // (an allocation failed half way: the tables that did grow stay grown, the frame takes the worst-case workspace)
    ctx->worst = ctx->run_cap == wc.run_cap && ctx->key_cap == wc.key_cap && ctx->slot_cap == wc.slot_cap && ctx->list_cap == (size_t)wc.list_cap &&
                 ctx->peak_cap == peak_worst && ctx->min_rho <= 1.0;
    ctx->n_cap_growths++;
    return true;
}

// the worst-case single-frame workspace behind a compact context (nullptr: ctx is worst-case itself, or no memory)
static lfdmi_ctx *get_spill(lfdmi_ctx *ctx) {
    if (ctx->worst) return nullptr;
    if (!ctx->spill) {
        lfdmi_caps wc;
        worst_caps(ctx->H, ctx->W, &wc);
        lfdmi_ctx *sp = nullptr;
        int rc = create_impl(ctx->device, ctx->H, ctx->W, 1, &wc, &sp);
        if (rc) {
            ctx->err = "worst-case workspace: " + (sp ? sp->err : std::string("creation failed"));
            if (sp) lfdmi_ctx_destroy(sp);
            return nullptr;
        }
        sp->timing = false;
        sp->scan_fused = false; // (one frame at a time, rarely: nothing to gain from the look-back scan)
        sp->scan_fused_cfg = false;
        ctx->spill = sp;
    }
    return ctx->spill;
}

extern "C" void lfdmi_ctx_destroy(lfdmi_ctx *ctx) {
    if (!ctx) return;
    if (ctx->spill) lfdmi_ctx_destroy(ctx->spill);
    hipSetDevice(ctx->device);
    if (ctx->stream) hipStreamSynchronize(ctx->stream);
    for (void *p : ctx->allocs) hipFree(p);
    if (ctx->stage) hipFree(ctx->stage);
    for (int i = 0; i < 2; i++) {
        if (ctx->feed_pin[i]) hipHostFree(ctx->feed_pin[i]);
        if (ctx->feed_dev[i]) hipFree(ctx->feed_dev[i]);
        if (ctx->feed_up[i]) hipEventDestroy(ctx->feed_up[i]);
    }
    if (ctx->feed_copy) hipStreamDestroy(ctx->feed_copy);
    if (ctx->feed_copy2) hipStreamDestroy(ctx->feed_copy2);
    if (ctx->feed_mid) hipEventDestroy(ctx->feed_mid);
    if (ctx->scratch) hipFree(ctx->scratch);
    if (ctx->rs_boxes) hipFree(ctx->rs_boxes);
    if (ctx->rs_sboxes) hipFree(ctx->rs_sboxes);
    if (ctx->rs_rowstart) hipFree(ctx->rs_rowstart);
    if (ctx->cat_dev) hipFree(ctx->cat_dev);
    for (auto e : ctx->ev_pool) hipEventDestroy(e);
    if (ctx->own_stream && ctx->stream) hipStreamDestroy(ctx->stream);
    for (int i = 0; i < 2; i++) {
        if (ctx->side[i]) hipStreamDestroy(ctx->side[i]);
        if (ctx->ev_join[i]) hipEventDestroy(ctx->ev_join[i]);
    }
    if (ctx->ev_fork) hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_rsfill) hipEventDestroy(ctx->ev_rsfill);
    if (ctx->res_host) hipHostFree(ctx->res_host);
    delete ctx;
}

extern "C" const char *lfdmi_last_error(lfdmi_ctx *ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }
extern "C" int lfdmi_max_inflight(lfdmi_ctx *ctx) { return ctx ? ctx->G : 0; }

extern "C" int lfdmi_set_stream(lfdmi_ctx *ctx, void *hip_stream) {
    if (!ctx) return LFDMI_ERR_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    if (ctx->own_stream && ctx->stream) { hipStreamSynchronize(ctx->stream); hipStreamDestroy(ctx->stream); }
    if (hip_stream) { ctx->stream = (hipStream_t)hip_stream; ctx->own_stream = false; }
    else { HIPCHK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)); ctx->own_stream = true; }
    return 0;
}

extern "C" int lfdmi_enable_timing(lfdmi_ctx *ctx, int on) {
    if (!ctx) return LFDMI_ERR_ARG;
    ctx->timing = on != 0;
    memset(ctx->t_ms, 0, sizeof ctx->t_ms);
    memset(ctx->t_n, 0, sizeof ctx->t_n);
    memset(ctx->t_units, 0, sizeof ctx->t_units);
    return 0;
}
extern "C" int lfdmi_timing_select(lfdmi_ctx *ctx, uint64_t mask) {
    static_assert(TG_COUNT <= 64, "one bit per timing slot");
    if (!ctx) return LFDMI_ERR_ARG;
    ctx->timing_mask = mask;
    return 0;
}
extern "C" int lfdmi_get_timing(lfdmi_ctx *ctx, float *ms, int32_t *launches, int64_t *units) {
    if (!ctx) return LFDMI_ERR_ARG;
    for (int i = 0; i < TG_COUNT; i++) {
        if (ms) ms[i] = ctx->t_ms[i];
        if (launches) launches[i] = ctx->t_n[i];
        if (units) units[i] = ctx->t_units[i];
    }
    return 0;
}
extern "C" int lfdmi_timing_slots(void) { return TG_COUNT; }
extern "C" const char *lfdmi_timing_name(int i) { return (i >= 0 && i < TG_COUNT) ? KID_NAMES[i] : ""; }

// ---- helpers ------------------------------------------------------------------------------
static int check_shape(lfdmi_ctx *ctx, int n, int h, int w) {
    if (!ctx) return LFDMI_ERR_ARG;
    if (n < 0 || h <= 0 || w <= 0) return fail(ctx, LFDMI_ERR_ARG, "bad shape");
    // both sides, not only the area: the band / tile tables (cellbm, candmask, fullbits, tile_list) are sized by max_h and
    // max_w separately, so a taller-but-narrower frame of the same area would index past them
    if (h > ctx->H || w > ctx->W || h > 8191 || w > 8191 || (LFD_WQ(w) * (size_t)h + 63) / 64 > SCAN_MAX_SEG)
        return fail(ctx, LFDMI_ERR_CAPACITY, "frame larger than the context was created for (height and width must both fit)");
    if (hipSetDevice(ctx->device) != hipSuccess) return fail(ctx, LFDMI_ERR_HIP, "hipSetDevice");
    (void)hipGetLastError(); // a failed earlier call must not poison this one
    ctx->last_h = h; ctx->last_w = w;
    ctx->stages_valid = true; // (the pass-level entry points overwrite this with what they keep)
    return 0;
}

static int ensure_scratch(lfdmi_ctx *ctx, size_t bytes) {
    if (ctx->scratch_bytes >= bytes) return 0;
    if (ctx->scratch) { HIPCHK(hipStreamSynchronize(ctx->stream)); HIPCHK(hipFree(ctx->scratch)); ctx->scratch = nullptr; ctx->scratch_bytes = 0; }
    HIPCHK(hipMalloc(&ctx->scratch, bytes));
    ctx->scratch_bytes = bytes;
    return 0;
}

static int ensure_stage(lfdmi_ctx *ctx, size_t bytes) {
    if (ctx->stage_bytes >= bytes) return 0;
    if (ctx->stage) { HIPCHK(hipStreamSynchronize(ctx->stream)); HIPCHK(hipFree(ctx->stage)); ctx->stage = nullptr; ctx->stage_bytes = 0; }
    HIPCHK(hipMalloc(&ctx->stage, bytes));
    ctx->stage_bytes = bytes;
    return 0;
}

static size_t dtype_size(int dtype) { return dtype == LFDMI_U8 ? 1 : ((dtype == LFDMI_F32 || dtype == LFDMI_F32_BE) ? 4 : 8); }

// returns a device pointer for `count` input bytes starting at src (+ offset), staging if needed
static int in_ptr(lfdmi_ctx *ctx, const void *src, size_t offset, size_t bytes, int loc, const void **out) {
    if (loc == LFDMI_DEVICE) { *out = (const char *)src + offset; return 0; }
    RET(ensure_stage(ctx, bytes));
    HIPCHK(hipMemcpyAsync(ctx->stage, (const char *)src + offset, bytes, hipMemcpyHostToDevice, ctx->stream));
    *out = ctx->stage;
    return 0;
}

static int out_copy(lfdmi_ctx *ctx, void *dst, size_t offset, const void *dev_src, size_t bytes, int loc) {
    if (!dst) return 0;
    HIPCHK(hipMemcpyAsync((char *)dst + offset, dev_src, bytes,
                          loc == LFDMI_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->stream));
    return 0;
}

// drain the stream; nc = frames of the chunk (stand-alone operators work on all of them)
static int sync(lfdmi_ctx *ctx, int nc = 0, const int *n_act = nullptr, const int *n_det = nullptr) {
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (ctx->timing) {
        int a[2] = {nc, 0}, d[2] = {nc, 0};
        collect_spans(ctx, n_act ? n_act : a, n_det ? n_det : d);
    }
    return 0;
}

// lfdmi_detect_batch, device-resident frames whose squares the sweep masked (rs_fold): the zero fill the caller is owed runs on
// side[1], started at stage k of the bright pass (LFDMI_RS_FILL_AT; -1: whatever is still pending), joined before the
// chunk's results are fetched.  It is 0.63 GB of partial-line stores and next to no instructions.
static int rs_fill_point(lfdmi_ctx *ctx, int k) {
    if (!ctx->rs_fill_frames) return 0;
    // the fill can go in two parts (half of the objects each) at two points of the step: rs_fill_at0 (first half; -1: no split) and
    // rs_fill_at (the rest); k < 0: whatever is still pending
    const bool first = k >= 0 && k == ctx->rs_fill_at0 && ctx->rs_fill_part == 0 && !ctx->rs_sorted;
    const bool rest = k < 0 || k == ctx->rs_fill_at;
    if (!first && !rest) return 0;
    float *fr = ctx->rs_fill_frames;
    static const int side_i = getenv("LFDMI_RS_FILL_SIDE") ? (atoi(getenv("LFDMI_RS_FILL_SIDE")) & 1) : 1; // developer
    hipStream_t sd = ctx->side[side_i];
    HIPCHK(hipEventRecord(ctx->ev_rsfill, ctx->stream));
    HIPCHK(hipStreamWaitEvent(sd, ctx->ev_rsfill, 0));
    {
        Span sp(ctx, KID_REMOVESTARS, 0, sd);
        if (ctx->rs_sorted)
            k_rs_fill_bands<<<dim3((ctx->rs_fill_h + RS_BAND_ROWS - 1) / RS_BAND_ROWS, ctx->rs_fill_nc), 256, 0, sd>>>(
                fr, ctx->rs_fill_h, ctx->rs_fill_w, ctx->rs_max_obj, ctx->rs_sboxes, ctx->rs_rowstart, ctx->rs_hmax);
        else {
            static const double frac = getenv("LFDMI_RS_FILL_FRAC") ? atof(getenv("LFDMI_RS_FILL_FRAC")) : 0.5; // developer: the first part's share of the objects
            const int nblk = (ctx->rs_max_obj + 3) / 4, half = std::max(0, std::min(nblk, (int)(nblk * frac)));
            const int b0 = ctx->rs_fill_part ? half : 0, b1 = first ? half : nblk;
            if (b1 > b0)
                k_rs_fill<<<dim3(b1 - b0, ctx->rs_fill_nc), 256, 0, sd>>>(fr, ctx->rs_fill_h, ctx->rs_fill_w, ctx->rs_max_obj,
                                                                          ctx->rs_count_dev, ctx->rs_boxes, b0);
        }
        KCHK("k_rs_fill");
    }
    HIPCHK(hipEventRecord(ctx->ev_rsfill, sd));
    ctx->rs_fill_inflight = true;
    if (first) ctx->rs_fill_part = 1;
    else { ctx->rs_fill_frames = nullptr; ctx->rs_fill_part = 0; }
    return 0;
}

static dim3 word_grid(int h, int w, int n) { return dim3((unsigned)((h * LFD_WQ(w) + 255) / 256), (unsigned)n); }

// ---- stage runners (device pointers only, nc <= G images in workspace slots 0..nc-1) --------
static int run_prep(lfdmi_ctx *ctx, const void *src, int dtype, int nc, int h, int w, int flip, int mode,
                    double minFlux, double addFlux, const int *active, bool zeroed = false, u64 *fullbits = nullptr,
                    const lfdmi_params *delta_dim = nullptr) {
    if (!zeroed) {
        HIPCHK(hipMemsetAsync(ctx->hist, 0, (size_t)nc * 256 * sizeof(int), ctx->stream));
        HIPCHK(hipMemsetAsync(ctx->cellbm, 0, (size_t)nc * ctx->bm_bands * CELLBM_WORDS * sizeof(u64), ctx->stream));
    }
    {
        Span sp(ctx, KID_PREP_HIST);
        static const int prep_rows = [] { // rows per workgroup (tuning knob): 1, 2, 4, 8 or 16
            const char *e = getenv("LFDMI_PREP_ROWS");
            int v = e ? atoi(e) : 8;
            return (v == 1 || v == 2 || v == 4 || v == 8 || v == 16) ? v : 8;
        }();
        const dim3 pgrid((h + prep_rows - 1) / prep_rows, nc);
#define LFD_PREP_LAUNCH(M_)                                                                                                          \
    k_prep_hist<M_><<<pgrid, 256, 0, ctx->stream>>>(src, dtype, h, w, flip, mode, minFlux, addFlux, ctx->gray, ctx->hist, ctx->cellbm, \
                                                    ctx->bm_bands, active, fullbits, prep_rows)
        if (delta_dim) { // the bright pass of lfdmi_detect_batch: the dim pass's values and histogram from the same sweep
#define LFD_PREP_DELTA(P_, RS_)                                                                                                       \
    k_prep_hist<1, true, P_, RS_><<<pgrid, 256, 0, ctx->stream>>>(src, dtype, h, w, flip, mode, minFlux, addFlux, ctx->gray, ctx->hist, \
                                                              ctx->cellbm, ctx->bm_bands, active, fullbits, prep_rows, ctx->dbits,    \
                                                              ctx->hist2, (float)delta_dim->minFlux, (float)delta_dim->addFlux, ctx->nzd, sky_fast, \
                                                              ctx->rs_sorted ? ctx->rs_sboxes : ctx->rs_boxes, ctx->rs_count_dev, ctx->rs_max_obj, \
                                                              ctx->rs_sorted ? ctx->rs_rowstart : nullptr, ctx->rs_hmax)
            // (all-sky shortcut of the sweep: exact when the smallest kept value already rounds to 1, see k_prep_hist)
            const bool sky_on = ctx->sky_fast; // (developer switch LFDMI_SKY_FAST)
            const float mfa = (float)delta_dim->minFlux + (float)delta_dim->addFlux;
            const int sky_fast = (sky_on && mode == LFDMI_PREP_BRIGHT && mfa > 0.5f) ? 1 : 0;
            const bool rs = ctx->rs_fold && w <= RS_MAXW && prep_rows <= RS_MAXROWS; // (remove_stars' squares masked in the sweep)
            if ((float)delta_dim->minFlux > 0.f) { if (rs) LFD_PREP_DELTA(true, true); else LFD_PREP_DELTA(true, false); }
            else { if (rs) LFD_PREP_DELTA(false, true); else LFD_PREP_DELTA(false, false); }
#undef LFD_PREP_DELTA
        } else
        if (dtype == LFDMI_F32 && (w & 3) == 0) { // float frames: the mode as a compile-time constant
            switch (mode & 3) {
            case 0: LFD_PREP_LAUNCH(0); break;
            case 1: LFD_PREP_LAUNCH(1); break;
            case 2: LFD_PREP_LAUNCH(2); break;
            default: LFD_PREP_LAUNCH(3); break;
            }
        } else LFD_PREP_LAUNCH(-1);
#undef LFD_PREP_LAUNCH
        KCHK("k_prep_hist");
    }
    Span sp(ctx, KID_LUT);
    k_lut<<<nc, 256, 0, ctx->stream>>>(ctx->hist, h * w, ctx->lut, active);
    KCHK("k_lut");
    return 0;
}

static bool all_ones(const uint8_t *k, int kh, int kw) {
    for (int i = 0; i < kh * kw; i++) if (!k[i]) return false;
    return true;
}

// dst = op(src) with optional LUT / bit rows; kernel mask on the host
static int run_morph(lfdmi_ctx *ctx, const uint8_t *src, uint8_t *dst, u64 *bits, const uint8_t *lut, const uint8_t *kernel,
                     int kh, int kw, int op, int nc, int h, int w, const int *active, const u64 *fullbits = nullptr, u64 *cellout = nullptr,
                     bool *marked = nullptr, bool sparse_fill = false) { // sparse_fill: dst is only read by the tile kernel through cellout's marks
    if (marked) *marked = false;
    if (!kernel || kh <= 0 || kw <= 0 || kh > LFDMI_MAX_MORPH_K || kw > LFDMI_MAX_MORPH_K)
        return fail(ctx, LFDMI_ERR_UNSUPPORTED, "structuring element must be 1..31 on both sides");
    Span sp(ctx, op ? KID_ERODE : KID_DILATE);
    if (all_ones(kernel, kh, kw) && (w % 16) == 0) {
        int IH = MORPH_TH + kh - 1;
        // staged tile + split horizontal result (2 dwords per 4 px) + output tile
        size_t lds = (size_t)IH * (MORPH_TW + 2 * MORPH_HALO) + (size_t)IH * MORPH_TW * 2 + (size_t)MORPH_TH * MORPH_TW;
        dim3 grid((w + MORPH_TW - 1) / MORPH_TW, (h + MORPH_TH - 1) / MORPH_TH, nc);
        const u64 *cand = nullptr;
        if (op == 1 && fullbits && kw >= 7 && !lut && !bits && grid.x <= 128 && kh <= 33) {
            // sparse input with its "full word" bits: zero fill + candidate tiles first, then only those tiles are eroded
            sparse_fill = sparse_fill && cellout && kh <= 9 && kw <= 33;
            k_erode_cand<<<dim3(grid.y, nc), 64, 0, ctx->stream>>>(fullbits, ctx->candmask, dst, h, w, kh, active, sparse_fill ? 0 : 1);
            KCHK("k_erode_cand");
            cand = ctx->candmask;
        }
        if (op == 0) k_morph_rect_v<0><<<grid, 256, lds, ctx->stream>>>(src, dst, bits, lut, h, w, kh, kw, active, cellout, ctx->bm_bands);
        else if (cand) { // a workgroup per tile row walks its candidate tiles
            dim3 rgrid(grid.y, nc);
            if (kh == 9 && kw == 9 && ctx->dc_specialize) // BASELINE configs[4]: unrolled, running minima by doubling
                k_morph_rect_rows<1, 9, 9><<<rgrid, 256, lds, ctx->stream>>>(src, dst, lut, h, w, kh, kw, active, cand, cellout, ctx->bm_bands);
            else k_morph_rect_rows<1><<<rgrid, 256, lds, ctx->stream>>>(src, dst, lut, h, w, kh, kw, active, cand, cellout, ctx->bm_bands);
        } else if (kh == 9 && kw == 9 && ctx->dc_specialize)
            k_morph_rect_v<1, 9, 9><<<grid, 256, lds, ctx->stream>>>(src, dst, bits, lut, h, w, kh, kw, active, cellout, ctx->bm_bands);
        else k_morph_rect_v<1><<<grid, 256, lds, ctx->stream>>>(src, dst, bits, lut, h, w, kh, kw, active, cellout, ctx->bm_bands);
        if (marked) *marked = cellout != nullptr;
        KCHK("k_morph_rect_v");
        if (cand && sparse_fill) {
            k_fill_around<<<dim3(grid.y, nc), 64, 0, ctx->stream>>>(cand, cellout, ctx->bm_bands, dst, h, w, active);
            KCHK("k_fill_around");
        }
    } else if (all_ones(kernel, kh, kw)) {
        int IH = MORPH_TH + kh - 1, IW = MORPH_TW + kw - 1;
        size_t lds = ((size_t)(IH * IW + 15) & ~(size_t)15) + (size_t)IH * MORPH_TW;
        dim3 grid((w + MORPH_TW - 1) / MORPH_TW, (h + MORPH_TH - 1) / MORPH_TH, nc);
        if (op == 0) k_morph_rect<0><<<grid, 256, lds, ctx->stream>>>(src, dst, bits, lut, h, w, kh, kw, active);
        else k_morph_rect<1><<<grid, 256, lds, ctx->stream>>>(src, dst, bits, lut, h, w, kh, kw, active);
        KCHK("k_morph_rect");
    } else {
        uint8_t *m = ctx->mask + (op ? LFDMI_MAX_MORPH_K * LFDMI_MAX_MORPH_K : 0);
        HIPCHK(hipMemcpyAsync(m, kernel, (size_t)kh * kw, hipMemcpyHostToDevice, ctx->stream));
        dim3 grid((w + 63) / 64, (h + 3) / 4, nc);
        if (op == 0) k_morph_generic<0><<<grid, 256, 0, ctx->stream>>>(src, dst, bits, lut, m, h, w, kh, kw, active);
        else k_morph_generic<1><<<grid, 256, 0, ctx->stream>>>(src, dst, bits, lut, m, h, w, kh, kw, active);
        KCHK("k_morph_generic");
    }
    return 0;
}

// Canny = NMS bit rows + hysteresis by run labelling; leaves edge bits in ctx->edgeb and the
// per-word run-count scan (compact run ids) + work lists + clearing of a sparse bit image: three wide kernels
static int run_scan(lfdmi_ctx *ctx, const u64 *bits, int val, int *scan, int cidx, int nc, int h, int w, int *wl_fg, int *wl_bg,
                    u64 *clear, const int *active) {
    int nseg = (h * LFD_WQ(w) + 63) / 64;
    dim3 grid((nseg + SCANW_WAVES * SCAN_SEGS - 1) / (SCANW_WAVES * SCAN_SEGS), nc);
    if (ctx->scan_fused && (int)grid.x <= SCAN_MAX_BLK) { // count + bases + write in one launch (k_scan_fused)
        if (++ctx->scan_epoch >= (1 << 22)) { // (the mark is 22 bits wide: start over on clean words)
            HIPCHK(hipMemsetAsync(ctx->scan_partial, 0, (size_t)ctx->G * SCAN_MAX_BLK * sizeof(u64), ctx->stream));
            ctx->scan_epoch = 1;
        }
        k_scan_fused<<<grid, 64 * SCANW_WAVES, 0, ctx->stream>>>(bits, val, ctx->scan_partial, ctx->scan_epoch, ctx->scan_spin, ctx->pass_flags, scan, h, w, wl_fg, wl_bg, clear,
                                                                  ctx->counters, cidx, ctx->run_cap, active);
        KCHK("k_scan_fused");
        return 0;
    }
    k_scan_count<<<grid, 64 * SCANW_WAVES, 0, ctx->stream>>>(bits, val, ctx->segcnt, h, w, wl_fg != nullptr, active);
    KCHK("k_scan_count");
    k_scan_bases<<<nc, SCAN_THREADS, 0, ctx->stream>>>(ctx->segcnt, ctx->counters, cidx, h, w, ctx->run_cap, wl_fg != nullptr, active);
    KCHK("k_scan_bases");
    k_scan_write<<<grid, 64 * SCANW_WAVES, 0, ctx->stream>>>(bits, val, ctx->segcnt, scan, h, w, wl_fg, wl_bg, clear, active);
    KCHK("k_scan_write");
    return 0;
}

// 8-connected labels of the surviving components in ctx->Lf
// want_keys: the contour stage follows (run_rects): k_frame_fg also makes the outer-border keys and their row extremes
static int run_canny(lfdmi_ctx *ctx, const uint8_t *img, int nc, int h, int w, double low_d, double high_d, const int *active,
                     bool nms_done = false, bool want_keys = false) {
    if (low_d > high_d) { double t = low_d; low_d = high_d; high_d = t; }
    int low = (int)floor(low_d), high = (int)floor(high_d);
    if (!nms_done) {
        Span sp(ctx, KID_CANNY_NMS);
        dim3 grid((w + CANNY_TW - 1) / CANNY_TW, (h + CANNY_TH - 1) / CANNY_TH, nc);
        if ((w % 16) == 0) k_canny_nms_v<<<grid, 256, 0, ctx->stream>>>(img, ctx->candb, ctx->strongb, h, w, low, high, active);
        else k_canny_nms<<<grid, 256, 0, ctx->stream>>>(img, ctx->candb, ctx->strongb, h, w, low, high, active);
        KCHK("k_canny_nms");
    }
    dim3 lg(WORDLIST_BLOCKS, nc);
    int rc = ctx->run_cap;
    { Span sp(ctx, KID_RUNS_INIT_FG);
      RET(run_scan(ctx, ctx->candb, 1, ctx->scanf_, C_NRUNF, nc, h, w, ctx->wl_fg, ctx->wl_bg, ctx->edgeb, active));
    }
    RET(rs_fill_point(ctx, 3 + 10 * ctx->cur_pass));
    if (ctx->frame_ccl) { // frames that fit the LDS tables; the rest (fallback flag) take the kernels below
        Span sp(ctx, KID_FRAME_FG);
        // labels + two bit sets, and (want_keys) as much again for the last-row table / the row slots of the outer borders:
        // everything the CU has when the label table is at its full size
        const bool keys = want_keys && ctx->fg_keys_on;
        size_t lds = (size_t)(ctx->frame_lds + 2 * (ctx->frame_lds / 32)) * sizeof(int);
        if (keys) lds = std::min<size_t>(2 * lds, 160 * 1024 - 1024);
        if (!keys) HIPCHK(hipMemsetAsync(ctx->fg_keys, 0, (size_t)nc * sizeof(int), ctx->stream));
        k_frame_fg<<<nc, FRAME_THREADS, lds, ctx->stream>>>(ctx->candb, ctx->strongb, ctx->scanf_, ctx->wl_fg, ctx->counters, ctx->Lf,
                                                            ctx->YMf, ctx->FLf, ctx->ROWf, ctx->edgeb, h, w, rc, ctx->frame_runcap, active, ctx->fb_fg,
                                                            ctx->pass_flags, (int)(lds / sizeof(int)), active ? ctx->perm_cur : nullptr, ctx->dc_profile ? nullptr : ctx->prof,
                                                            keys ? ctx->keys : nullptr, ctx->bigkeys, ctx->medkeys, ctx->rowext, ctx->key_cap, ctx->slot_cap,
                                                            keys ? ctx->fg_keys : nullptr, (ctx->frame_dbg & 16) ? 1 : 0);
        KCHK("k_frame_fg");
        if (!ctx->general_on) return 0; // (a frame that did not fit raises PASS_FLAG_GENERAL: the caller runs the chunk again)
        active = ctx->fb_fg;
    }
    { Span sp(ctx, KID_RUNS_INIT_FG);
      k_runs_init<<<lg, 256, 0, ctx->stream>>>(ctx->candb, 1, ctx->scanf_, ctx->Lf, ctx->YMf, ctx->FLf, ctx->ROWf, h, w, rc,
                                               ctx->wl_fg, ctx->counters, C_NFGW, active);
      KCHK("k_runs_init"); }
    { Span sp(ctx, KID_RUNS_MERGE8);
      k_runs_merge8<<<lg, 256, 0, ctx->stream>>>(ctx->candb, ctx->scanf_, ctx->Lf, h, w, rc, ctx->wl_fg, ctx->counters, C_NFGW, active);
      KCHK("k_runs_merge8"); }
    { Span sp(ctx, KID_RUNS_FLATTEN_FG);
      k_runs_flatten<<<lg, 256, 0, ctx->stream>>>(ctx->candb, 1, ctx->strongb, ctx->scanf_, ctx->Lf, ctx->YMf, ctx->FLf, h, w, rc,
                                                  ctx->wl_fg, ctx->counters, C_NFGW, active);
      KCHK("k_runs_flatten"); }
    { Span sp(ctx, KID_EDGE);
      k_edge_from_cand<<<lg, 256, 0, ctx->stream>>>(ctx->candb, ctx->scanf_, ctx->Lf, ctx->FLf, ctx->edgeb, h, w, rc, ctx->wl_fg,
                                                    ctx->counters, C_NFGW, active);
      KCHK("k_edge_from_cand"); }
    return 0;
}

// dilate (all-ones kernel) + Canny NMS in one tile kernel when the shapes allow; returns false otherwise
static bool can_fuse_dilate_canny(const uint8_t *kernel, int kh, int kw, int w) {
    if (!kernel || kh <= 0 || kw <= 0 || kh > LFDMI_MAX_MORPH_K || kw > LFDMI_MAX_MORPH_K || (w % 16) != 0) return false;
    if (!all_ones(kernel, kh, kw)) return false;
    int ax = kw / 2;
    return ax + 2 <= CANNY_HALO - 4 + 4 && (kw - 1 - ax) + 2 <= CANNY_HALO - 4 + 4 && ax + 4 <= CANNY_HALO && (kw - 1 - ax) + 4 <= CANNY_HALO;
}

static int run_dilate_canny(lfdmi_ctx *ctx, const uint8_t *src, int nc, int h, int w, const uint8_t *lut, int kh, int kw,
                            const int *active, const u64 *cellbm) {
    const bool use_bm = cellbm != nullptr;
    if (kh <= DCW_MAXKH) { // one wave per 64 x 16 tile, mask-driven (the sparse pass images)
        int IH = DCW_PH + kh - 1, MGB = DCW_MH * CANNY_MW * 2;
        size_t lds = (size_t)(IH * DCW_TS > MGB ? IH * DCW_TS : MGB) + (size_t)IH * DCW_NWD * 4 + (size_t)DCW_PH * DCW_TS;
        int tiles_x = (w + CANNY_TW - 1) / CANNY_TW, tiles_y = (h + DCW_TH - 1) / DCW_TH;
        if (ctx->dc_tilelist && tiles_x <= 128 && tiles_y <= DCT_MAXBANDS) {
            // active-tile list from the cell bitmap (or every tile), background of the bit planes, then the tile stages
            Span sp(ctx, KID_DILATE_CANNY);
            // (one workgroup per frame writing 1-6 MB of background is bound by its own store latency: the fill is spread over
            // a workgroup per ~256 KB)
            int fill_parts = (int)std::min<size_t>(64, std::max<size_t>(1, ((size_t)h * LFD_WQ(w) * 24 + (256u << 10) - 1) / (256u << 10)));
            if (const char *e = getenv("LFDMI_DCT_FILLPARTS")) fill_parts = std::max(1, atoi(e));
            k_dc_tiles<<<dim3(nc, fill_parts), DCT_THREADS, 0, ctx->stream>>>(cellbm, ctx->bm_bands, lut, ctx->tile_list, ctx->tile_cap,
                                                             ctx->counters, ctx->equb, ctx->candb, ctx->strongb, ctx->keep_equ ? ctx->equ : nullptr,
                                                             h, w, active, active ? ctx->perm_cur : nullptr);
            KCHK("k_dc_tiles");
            const int *tperm = active ? ctx->perm_cur : nullptr;
            if (ctx->use_tile_perm && ctx->use_perm && nc <= 1024 && nc > 8) { // frames sorted by work, dealt evenly to the XCDs
                k_tile_perm<<<1, 1024, 0, ctx->stream>>>(active, ctx->counters, nc, ctx->perm_tiles);
                KCHK("k_tile_perm");
                tperm = ctx->perm_tiles;
            }
            // many short waves: the dispatcher evens out frames and regions with more occupied tiles than others (a wave gets
            // 1 / parts of the frame's list: 3-5 tiles on SDSS frames; 47 parts: 1.46 ms per step, 256: 1.29 ms)
            int parts = ctx->dc_parts > 0 ? ctx->dc_parts : std::max(8, std::min(256, tiles_x * tiles_y / 8));
            unsigned grid = 8u * ((nc + 7) / 8) * parts;
            long long *prof = nullptr;
            if (ctx->dc_profile && ctx->prof) {
                HIPCHK(hipMemsetAsync(ctx->prof, 0, (size_t)nc * 8 * sizeof(long long), ctx->stream));
                prof = ctx->prof;
            }
#define LFD_DCT_LAUNCH(PROF_, KH_, KW_)                                                                                              \
    k_dilate_canny_t<PROF_, KH_, KW_><<<grid, 64, lds, ctx->stream>>>(src, ctx->keep_equ ? ctx->equ : nullptr, ctx->equb, ctx->candb, \
                                                                      ctx->strongb, lut, h, w, kh, kw, 0, 255, active, nc, parts,     \
                                                                      ctx->tile_list, ctx->tile_cap, ctx->counters, prof, tperm)
            const bool spec = ctx->dc_specialize;
            if (prof) {
                if (spec && kh == 4 && kw == 4) LFD_DCT_LAUNCH(true, 4, 4);
                else if (spec && kh == 9 && kw == 9) LFD_DCT_LAUNCH(true, 9, 9);
                else LFD_DCT_LAUNCH(true, 0, 0);
            } else {
                if (spec && kh == 4 && kw == 4) LFD_DCT_LAUNCH(false, 4, 4);
                else if (spec && kh == 9 && kw == 9) LFD_DCT_LAUNCH(false, 9, 9);
                else LFD_DCT_LAUNCH(false, 0, 0);
            }
#undef LFD_DCT_LAUNCH
            KCHK("k_dilate_canny_t");
            return 0;
        }
        int S = ctx->dc_strip, SS = ctx->dc_substrips, nstripx = (tiles_x + S * SS - 1) / (S * SS);
        unsigned grid = 8u * ((nc + 7) / 8) * tiles_y * nstripx; // frame = 8 * (j / strips) + (block & 7): one XCD per frame
        Span sp(ctx, KID_DILATE_CANNY);
        k_dilate_canny_w<<<grid, 64, lds, ctx->stream>>>(src, ctx->keep_equ ? ctx->equ : nullptr, ctx->equb, ctx->candb,
                                                          ctx->strongb, lut, h, w, kh, kw, 0, 255, active, nc, tiles_x, nstripx, S, SS,
                                                          cellbm, ctx->bm_bands);
        (void)use_bm;
        KCHK("k_dilate_canny_w");
        return 0;
    }
    int IH = CANNY_TH + 4 + kh - 1;
    size_t a = (size_t)IH * (CANNY_TW + 2 * CANNY_HALO) + (size_t)IH * (CANNY_MW / 4) * 8;
    size_t b = (size_t)2 * (CANNY_TH + 2) * CANNY_MW * 4;
    size_t lds = a > b ? a : b;
    dim3 grid((w + CANNY_TW - 1) / CANNY_TW, (h + CANNY_TH - 1) / CANNY_TH, nc);
    Span sp(ctx, KID_DILATE_CANNY);
    k_dilate_canny_v<<<grid, 256, lds, ctx->stream>>>(src, ctx->keep_equ ? ctx->equ : nullptr, ctx->equb, ctx->candb, ctx->strongb,
                                                       lut, h, w, kh, kw, 0, 255, active);
    KCHK("k_dilate_canny_v");
    return 0;
}

// cv2.getGaussianKernel(n, sigma, CV_32F)
static int gauss_taps(lfdmi_ctx *ctx, int n, double sigma, GaussTaps *t) {
    static const float small_tab[4][7] = {{1.f}, {0.25f, 0.5f, 0.25f}, {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f},
                                          {0.03125f, 0.109375f, 0.21875f, 0.28125f, 0.21875f, 0.109375f, 0.03125f}};
    if (n <= 0 || n > 31 || (n & 1) == 0) return fail(ctx, LFDMI_ERR_ARG, "gaussKernel must be odd, 1..31");
    const float *fixed = (n <= 7 && sigma <= 0) ? small_tab[n >> 1] : nullptr;
    double sigmaX = sigma > 0 ? sigma : ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
    double scale2X = -0.5 / (sigmaX * sigmaX), sum = 0;
    for (int i = 0; i < n; i++) {
        double x = i - (n - 1) * 0.5;
        double v = fixed ? (double)fixed[i] : exp(scale2X * x * x);
        t->k[i] = (float)v;
        sum += t->k[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < n; i++) t->k[i] = (float)(t->k[i] * sum);
    t->n = n;
    return 0;
}

static int run_gauss(lfdmi_ctx *ctx, const uint8_t *src, uint8_t *dst, int nc, int h, int w, int ksize, double sigma, const int *active) {
    GaussTaps t;
    RET(gauss_taps(ctx, ksize, sigma, &t));
    int r = ksize / 2, IW = GAUSS_TW + 2 * r, IH = GAUSS_TH + 2 * r;
    size_t lds = (size_t)((IH * IW + 15) & ~15) + (size_t)IH * GAUSS_TW * sizeof(float);
    Span sp(ctx, KID_MISC);
    k_gauss<<<dim3((w + GAUSS_TW - 1) / GAUSS_TW, (h + GAUSS_TH - 1) / GAUSS_TH, nc), 256, lds, ctx->stream>>>(src, dst, h, w, t, active);
    KCHK("k_gauss");
    return 0;
}

static int zero_counters(lfdmi_ctx *ctx, int nc) {
    HIPCHK(hipMemsetAsync(ctx->counters, 0, (size_t)nc * C_COUNT * sizeof(int), ctx->stream));
    return 0;
}

// contours -> rectangles -> box bit rows (needs run_canny's outputs)
static int run_rects(lfdmi_ctx *ctx, int nc, int h, int w, int mode, int method, double minLen, double lwTresh, const int *active) {
    if (method != LFDMI_CHAIN_APPROX_NONE && method != LFDMI_CHAIN_APPROX_SIMPLE)
        return fail(ctx, LFDMI_ERR_UNSUPPORTED, "contoursMethod: only CHAIN_APPROX_NONE / CHAIN_APPROX_SIMPLE");
    if (mode != LFDMI_RETR_EXTERNAL && mode != LFDMI_RETR_LIST && mode != LFDMI_RETR_CCOMP && mode != LFDMI_RETR_TREE)
        return fail(ctx, LFDMI_ERR_UNSUPPORTED, "contoursMode: RETR_EXTERNAL / RETR_LIST / RETR_CCOMP / RETR_TREE");
    dim3 lg(WORDLIST_BLOCKS, nc);
    int rc = ctx->run_cap;
    { Span sp(ctx, KID_RUNS_INIT_BG);
      RET(run_scan(ctx, ctx->edgeb, 0, ctx->scanb_, C_NRUNB, nc, h, w, nullptr, nullptr, ctx->boxb, active)); }
    RunTabs rt;
    rt.cand = ctx->candb; rt.edge = ctx->edgeb; rt.scanf = ctx->scanf_; rt.scanb = ctx->scanb_;
    rt.Lf = ctx->Lf; rt.YMf = ctx->YMf; rt.SBf = ctx->SBf; rt.ROWf = ctx->ROWf; rt.FLf = ctx->FLf;
    rt.Lb = ctx->Lb; rt.YMb = ctx->YMb; rt.FLb = ctx->FLb; rt.SBb = ctx->SBb; rt.PAb = ctx->PAb; rt.ROWb = ctx->ROWb;
    rt.run_cap = rc;
    const int *gen = active; // frames for the general run kernels
    RET(rs_fill_point(ctx, 4 + 10 * ctx->cur_pass));
    if (ctx->frame_ccl) {
        Span sp(ctx, KID_FRAME_BG);
        size_t lds = (size_t)(ctx->frame_lds + 4 * (ctx->frame_lds / 32)) * sizeof(int);
        k_frame_contours<<<nc, FRAME_THREADS, lds, ctx->stream>>>(rt, ctx->wl_fg, ctx->wl_bg, ctx->counters, ctx->keys, ctx->bigkeys,
                                                                  ctx->medkeys, ctx->rowext, ctx->rsa, h, w, ctx->key_cap, ctx->slot_cap,
                                                                  ctx->frame_runcap, active, ctx->fb_bg, ctx->pass_flags, ctx->dc_profile ? nullptr : ctx->prof,
                                                                  ctx->frame_lds, active ? ctx->perm_cur : nullptr, ctx->frame_dbg, ctx->fg_keys);
        KCHK("k_frame_contours");
        gen = ctx->fb_bg;
    }
    if (!ctx->frame_ccl || ctx->general_on) {
    { Span sp(ctx, KID_RUNS_INIT_BG);
      k_runs_init<<<lg, 256, 0, ctx->stream>>>(ctx->edgeb, 0, ctx->scanb_, ctx->Lb, ctx->YMb, ctx->FLb, ctx->ROWb, h, w, rc,
                                               ctx->wl_bg, ctx->counters, C_NBGW, gen);
      KCHK("k_runs_init(bg)"); }
    { Span sp(ctx, KID_RUNS_MERGE4);
      k_runs_merge4_bg<<<lg, 256, 0, ctx->stream>>>(ctx->edgeb, ctx->scanb_, ctx->Lb, h, w, rc, ctx->wl_bg, ctx->counters, C_NBGW, gen);
      KCHK("k_runs_merge4_bg"); }
    { Span sp(ctx, KID_RUNS_FLATTEN_BG);
      k_runs_flatten<<<lg, 256, 0, ctx->stream>>>(ctx->edgeb, 0, nullptr, ctx->scanb_, ctx->Lb, ctx->YMb, ctx->FLb, h, w, rc,
                                                  ctx->wl_bg, ctx->counters, C_NBGW, gen);
      KCHK("k_runs_flatten(bg)");
      k_bg_extent<<<lg, 256, 0, ctx->stream>>>(ctx->edgeb, ctx->scanb_, ctx->Lb, ctx->YMb, ctx->FLb, h, w, rc, ctx->wl_bg,
                                               ctx->counters, C_NBGW, gen);
      KCHK("k_bg_extent"); }
    { Span sp(ctx, KID_KEYS);
    k_keys<<<lg, 256, 0, ctx->stream>>>(rt, ctx->keys, ctx->bigkeys, ctx->medkeys, ctx->rowext, ctx->counters, h, w, ctx->key_cap, ctx->slot_cap,
                                         ctx->wl_fg, ctx->wl_bg, gen);
    KCHK("k_keys"); }
    { Span sp(ctx, KID_EXTREMES);
    k_extremes<<<lg, 256, 0, ctx->stream>>>(rt, ctx->rowext, h, w, ctx->slot_cap, ctx->wl_fg, ctx->counters, gen);
    KCHK("k_extremes"); }
    }
    if (mode == LFDMI_RETR_EXTERNAL) { // hole borders and enclosed components drop out before the rectangles
        Span sp(ctx, KID_KEYS);
        k_filter_external<<<dim3(8, nc), 256, 0, ctx->stream>>>(rt, ctx->keys, ctx->rowext, ctx->counters, h, w, ctx->key_cap, ctx->slot_cap, active);
        KCHK("k_filter_external");
    }
    { Span sp(ctx, KID_RECTS);
    // Three independent kernels by key height (short: a lane per key; medium / tall: a wave per key).
    // Each is a handful of long serial hulls, so they run side by side: fork two helper streams off
    // the launch stream, join before the fill.
    int cap = h + 2; // rows a key can span (a hole border adds one row above and below)
    // (k_rects: 8 workgroups of 64 lanes per frame -- sky frames have ~500 short keys; 32 workgroups, most of them without a key
    // but each holding 24 KB of LDS, kept the two wave-per-key kernels on the side streams waiting: 0.405 -> 0.355 ms per step)
    static const int rects_grid = getenv("LFDMI_RECTS_GRID") ? std::max(1, atoi(getenv("LFDMI_RECTS_GRID"))) : 8; // developer knobs
    static const bool rects_side = getenv("LFDMI_RECTS_SIDE") ? atoi(getenv("LFDMI_RECTS_SIDE")) != 0 : true;
    const int wave_prep = ctx->rects_prep ? 1 : 0; // (developer switch LFDMI_RECTS_PREP)
    static const int med_grid = getenv("LFDMI_RECTS_MED_GRID") ? std::max(1, atoi(getenv("LFDMI_RECTS_MED_GRID"))) : 64;
    static const int big_grid = getenv("LFDMI_RECTS_BIG_GRID") ? std::max(1, atoi(getenv("LFDMI_RECTS_BIG_GRID"))) : 8;
    if (!rects_side) { // all three on the launch stream, one after the other (to see what the side streams buy)
        k_rects_big<<<dim3(med_grid, nc), 64, (size_t)BIG_KEY_ROWS * 4 * sizeof(int2), ctx->stream>>>(
            ctx->keys, ctx->medkeys, C_NMED, ctx->rowext, ctx->quads, ctx->counters, h, w, ctx->key_cap, ctx->slot_cap, BIG_KEY_ROWS,
            minLen, lwTresh, active, wave_prep, 0, 0);
        k_rects_big<<<dim3(big_grid, nc), 64, (size_t)cap * 4 * sizeof(int2), ctx->stream>>>(
            ctx->keys, ctx->bigkeys, C_NBIG, ctx->rowext, ctx->quads, ctx->counters, h, w, ctx->key_cap, ctx->slot_cap, cap,
            minLen, lwTresh, active, wave_prep, 0, 0);
        k_rects<<<dim3(rects_grid, nc), 64, 0, ctx->stream>>>(ctx->keys, ctx->rowext, nullptr, ctx->quads, ctx->counters, h, w,
                                                        ctx->key_cap, ctx->slot_cap, minLen, lwTresh, active);
        KCHK("k_rects");
    } else {
    HIPCHK(hipEventRecord(ctx->ev_fork, ctx->stream));
    HIPCHK(hipStreamWaitEvent(ctx->side[0], ctx->ev_fork, 0));
    k_rects_big<<<dim3(med_grid, nc), 64, (size_t)BIG_KEY_ROWS * 4 * sizeof(int2), ctx->side[0]>>>(
        ctx->keys, ctx->medkeys, C_NMED, ctx->rowext, ctx->quads, ctx->counters, h, w, ctx->key_cap, ctx->slot_cap, BIG_KEY_ROWS,
        minLen, lwTresh, active, wave_prep, 0, 0);
    KCHK("k_rects_med");
    HIPCHK(hipEventRecord(ctx->ev_join[0], ctx->side[0]));
    HIPCHK(hipStreamWaitEvent(ctx->side[1], ctx->ev_fork, 0));
    // the tall keys (more than BIG_KEY_ROWS rows) in two launches over one list: the few that are taller than tall_mid rows with the
    // full-height LDS footprint (48 KB for one wave: three per CU), the others with a tenth of it and four times the workgroups
    static const int tall_mid = getenv("LFDMI_RECTS_TALL_MID") ? std::max(0, atoi(getenv("LFDMI_RECTS_TALL_MID"))) : 320; // (0: one launch)
    static const int tall_grid = getenv("LFDMI_RECTS_TALL_GRID") ? std::max(1, atoi(getenv("LFDMI_RECTS_TALL_GRID"))) : 32;
    if (tall_mid > BIG_KEY_ROWS && tall_mid < cap) {
        k_rects_big<<<dim3(big_grid, nc), 64, (size_t)cap * 4 * sizeof(int2), ctx->side[1]>>>(
            ctx->keys, ctx->bigkeys, C_NBIG, ctx->rowext, ctx->quads, ctx->counters, h, w, ctx->key_cap, ctx->slot_cap, cap,
            minLen, lwTresh, active, wave_prep, tall_mid, 0);
        k_rects_big<<<dim3(tall_grid, nc), 64, (size_t)tall_mid * 4 * sizeof(int2), ctx->side[1]>>>(
            ctx->keys, ctx->bigkeys, C_NBIG, ctx->rowext, ctx->quads, ctx->counters, h, w, ctx->key_cap, ctx->slot_cap, tall_mid,
            minLen, lwTresh, active, wave_prep, 0, 1);
    } else
    k_rects_big<<<dim3(big_grid, nc), 64, (size_t)cap * 4 * sizeof(int2), ctx->side[1]>>>(
        ctx->keys, ctx->bigkeys, C_NBIG, ctx->rowext, ctx->quads, ctx->counters, h, w, ctx->key_cap, ctx->slot_cap, cap,
        minLen, lwTresh, active, wave_prep, 0, 0);
    KCHK("k_rects_big");
    HIPCHK(hipEventRecord(ctx->ev_join[1], ctx->side[1]));
    k_rects<<<dim3(rects_grid, nc), 64, 0, ctx->stream>>>(ctx->keys, ctx->rowext, nullptr, ctx->quads, ctx->counters, h, w,
                                                    ctx->key_cap, ctx->slot_cap, minLen, lwTresh, active);
    KCHK("k_rects");
    HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_join[0], 0));
    HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_join[1], 0));
    RET(rs_fill_point(ctx, 5 + 10 * ctx->cur_pass));
    }
    }
    Span sp(ctx, KID_FILL);
    k_fill_quads<<<dim3(FILL_BLOCKS, nc), 256, 0, ctx->stream>>>(ctx->quads, ctx->counters, ctx->boxb, h, w, ctx->key_cap, active);
    KCHK("k_fill_quads");
    return 0;
}

// does this context's accumulator / peak storage hold HoughLines(rho, theta) on an h x w image?
static bool hough_fits(const lfdmi_ctx *ctx, int h, int w, double rho_d, double theta_d) {
    int na, nr;
    lfdmi_hough_dims(h, w, rho_d, theta_d, &na, &nr);
    return na > 0 && nr > 0 && na <= MAX_ANGLES && (size_t)(na + 2) * (nr + 2) <= ctx->acc_cap && (ctx->worst ? (size_t)na * nr <= ctx->peak_cap : true);
}

static int ensure_tables(lfdmi_ctx *ctx, int h, int w, double rho_d, double theta_d) {
    for (int i = 0; i < LFDMI_MAX_SCALES; i++) {
        auto &t = ctx->tabs[i];
        if (t.rho == rho_d && t.theta == theta_d && t.h == h && t.w == w) {
            t.stamp = ++ctx->tab_clock;
            ctx->tab_cur = i; ctx->numangle = t.numangle; ctx->numrho = t.numrho;
            return 0;
        }
    }
    float rho = (float)rho_d, theta = (float)theta_d;
    if (!(rho > 0) || !(theta > 0)) return fail(ctx, LFDMI_ERR_ARG, "rho and theta must be positive");
    int na, nr;
    lfdmi_hough_dims(h, w, rho_d, theta_d, &na, &nr);
    if (na <= 0 || nr <= 0 || na > MAX_ANGLES || (size_t)(na + 2) * (nr + 2) > ctx->acc_cap)
        return fail(ctx, LFDMI_ERR_CAPACITY, "Hough accumulator larger than the workspace (rho < 1 px or theta < 1 deg)");
    int slot = 0;
    for (int i = 1; i < LFDMI_MAX_SCALES; i++) if (ctx->tabs[i].stamp < ctx->tabs[slot].stamp) slot = i; // least recently used
    auto &t = ctx->tabs[slot];
    // the upload below reads t.host asynchronously (pageable memory: staged before the call returns) and is ordered on
    // the launch stream behind every kernel that still reads the entry's old table
    t.host.assign((size_t)2 * na, 0.f);
    float irho = 1 / rho;
    float ang = 0.f;
    for (int n = 0; n < na; ang += theta, n++) { // createTrigTable: float accumulation of the angle
        t.host[n] = (float)(cos((double)ang) * irho);
        t.host[na + n] = (float)(sin((double)ang) * irho);
    }
    HIPCHK(hipMemcpyAsync(ctx->tab + (size_t)slot * 2 * MAX_ANGLES, t.host.data(), t.host.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    t.rho = rho_d; t.theta = theta_d; t.h = h; t.w = w; t.numangle = na; t.numrho = nr;
    t.stamp = ++ctx->tab_clock;
    ctx->tab_cur = slot; ctx->numangle = na; ctx->numrho = nr;
    return 0;
}

// HoughLines on equ bits (image 0) and optionally box bits (image 1); leaves accumulators and
// unsorted peak keys in the workspace; top-K lines in ctx->lines when K > 0
static int run_hough(lfdmi_ctx *ctx, int nc, int h, int w, double rho, double theta, int threshold, int n_img, int K,
                     int need_detect, const int *active) {
    RET(ensure_tables(ctx, h, w, rho, theta));
    int na = ctx->numangle, nr = ctx->numrho;
    // Angles per workgroup (AW, a power of two <= 64) and the accumulator rows each angle slab can reach:
    // r = x c + y s over the image rectangle (+- 2 bins of slack).  The LDS slab of a workgroup holds only those rows,
    // AW votes wide; the largest AW whose widest slab fits in LDS is used (SDSS, rho 20: 64 angles x <= 128 rows;
    // 4096 x 4096, rho 5: 16 angles x ~1 160 rows).
    const std::vector<float> &tab_host = ctx->tabs[ctx->tab_cur].host;
    const int half = (nr - 1) / 2;
    VoteRanges rng;
    int aw_log2 = 6, nslabs = 1, nbmax = nr;
    for (;; aw_log2--) {
        int AW = 1 << aw_log2;
        nslabs = (na + AW - 1) / AW;
        nbmax = 1;
        for (int sl = 0; sl < VOTE_MAX_SLABS; sl++) { rng.lo[sl] = -half; rng.hi[sl] = nr - 1 - half; }
        if (nslabs <= VOTE_MAX_SLABS && (int)tab_host.size() == 2 * na) {
            for (int sl = 0; sl < nslabs; sl++) {
                double rmin = 0, rmax = 0;
                for (int n = sl * AW; n < na && n < (sl + 1) * AW; n++) {
                    double c = tab_host[n], sn = tab_host[na + n];
                    for (int cx = 0; cx < 2; cx++)
                        for (int cy = 0; cy < 2; cy++) {
                            double r = (cx ? w : 0) * c + (cy ? h : 0) * sn;
                            rmin = r < rmin ? r : rmin;
                            rmax = r > rmax ? r : rmax;
                        }
                }
                int lo = (int)floor(rmin) - 2, hi = (int)ceil(rmax) + 2;
                rng.lo[sl] = lo < -half ? -half : lo;
                rng.hi[sl] = hi > nr - 1 - half ? nr - 1 - half : hi;
                nbmax = std::max(nbmax, rng.hi[sl] - rng.lo[sl] + 1);
            }
        } else nbmax = nr;
        if (((size_t)nbmax << aw_log2) * 4 + 256 <= 152 * 1024 || aw_log2 == 0) break;
    }
    if (((size_t)nbmax << aw_log2) * 4 + 256 > 152 * 1024) return fail(ctx, LFDMI_ERR_CAPACITY, "numrho too large for LDS");
    // Chunk length per slab: (len - 1) * max |cos / rho| over the slab's angles stays below one bin (k_hough_vote's two-bin
    // split; its exact checks do not depend on this bound, only its speed does).  Class A: the length every slab can take;
    // class B: the slabs that can take at least twice that (angles away from the horizontal), with their own, longer cut.
    int cm_a = CHUNK_MAX, cm_b = 0;
    rng.cls_b = 0u;
    {
        const int AW = 1 << aw_log2;
        int lmax[VOTE_MAX_SLABS];
        const bool per_slab = nslabs <= VOTE_MAX_SLABS && (int)tab_host.size() == 2 * na;
        for (int sl = 0; sl < (per_slab ? nslabs : 1); sl++) {
            double cmax = 0;
            for (int n = per_slab ? sl * AW : 0; n < na && (!per_slab || n < (sl + 1) * AW); n++) cmax = std::max(cmax, (double)fabsf(tab_host[n]));
            double l = cmax > 0 ? floor(0.999 / cmax) + 1 : CHUNK_MAX;
            lmax[sl] = (int)std::min<double>(CHUNK_MAX, std::max<double>(1, l));
            cm_a = std::min(cm_a, lmax[sl]);
        }
        const bool use_b = ctx->vote_classes; // (developer switch LFDMI_VOTE_CLASSES)
        if (per_slab && use_b) {
            int mb = CHUNK_MAX + 1;
            for (int sl = 0; sl < nslabs; sl++)
                if (lmax[sl] >= 2 * cm_a) { rng.cls_b |= 1u << sl; mb = std::min(mb, lmax[sl]); }
            if (rng.cls_b) cm_b = mb;
        }
    }
    dim3 wg = word_grid(h, w, nc);
    {
        // cut each pixel list into pieces so that a launch carries several workgroups per CU
        int nsplit = 1;
        static const int vote_wgs = getenv("LFDMI_VOTE_WGS") ? atoi(getenv("LFDMI_VOTE_WGS")) : 6144;
        while (nsplit < ctx->vote_split && nslabs * n_img * nc * nsplit < vote_wgs) nsplit <<= 1;
        // two images: their pieces come out of one pool per slab, shared out by list length on the device (k_hough_vote: balance)
        const bool vote_balance = ctx->vote_balance; // (developer switch LFDMI_VOTE_BALANCE)
        // (unsplit lists -- 4096 x 4096 frames at rho = 5: 12 slabs x 2 images x 256 frames fill the GPU already -- gain nothing from
        // four pieces per slab: 2.78 vs 2.79 ms)
        const int balance = (vote_balance && n_img == 2 && nsplit >= 2) ? 1 : 0;
        int acc_n = (na + 2) * (nr + 2);
        { Span sp(ctx, KID_PIXLIST, need_detect);
        // per-slot accumulator pairs are 2 * acc_cap apart; the kernel indexes by slot itself
        k_pixlist<<<dim3((wg.x + PIXLIST_WORDS - 1) / PIXLIST_WORDS, wg.y, n_img), 256, 0, ctx->stream>>>(ctx->equb, ctx->boxb, ctx->pix_equ, ctx->pix_box, ctx->counters, cm_a, cm_b,
                                                                    h, w, ctx->list_cap, (nsplit > 1 || balance) ? ctx->accum : nullptr, acc_n, ctx->acc_cap,
                                                                    active, need_detect);
        KCHK("k_pixlist"); }
        Span sp(ctx, KID_VOTE, need_detect);
        const int vsplit = balance ? 2 * nsplit : nsplit;
        dim3 vgrid(nslabs * vsplit, balance ? 1 : n_img, nc);
        size_t vlds = ((size_t)nbmax << aw_log2) * 4 + 256;
        static const int vote_threads = getenv("LFDMI_VOTE_THREADS") ? std::max(64, std::min(1024, atoi(getenv("LFDMI_VOTE_THREADS")) & ~63)) : VOTE_THREADS; // developer knob
#define LFD_LAUNCH_VOTE(L)                                                                                          \
    case L:                                                                                                         \
        k_hough_vote<L><<<vgrid, vote_threads, vlds, ctx->stream>>>(ctx->pix_equ, ctx->pix_box, ctx->counters,           \
                                                                   ctx->tab + (size_t)ctx->tab_cur * 2 * MAX_ANGLES, \
                                                                   ctx->accum, na, nr, vsplit, ctx->list_cap,       \
                                                                   ctx->acc_cap, active, need_detect, rng, balance); \
        break;
        switch (aw_log2) {
            LFD_LAUNCH_VOTE(6) LFD_LAUNCH_VOTE(5) LFD_LAUNCH_VOTE(4) LFD_LAUNCH_VOTE(3) LFD_LAUNCH_VOTE(2)
            LFD_LAUNCH_VOTE(1) LFD_LAUNCH_VOTE(0)
        }
#undef LFD_LAUNCH_VOTE
        KCHK("k_hough_vote");
    }
    { Span sp(ctx, KID_PEAKS, need_detect);
    static const int peak_rows = getenv("LFDMI_PEAK_ROWS") ? std::max(8, atoi(getenv("LFDMI_PEAK_ROWS"))) : 32; // bins per workgroup at least
    k_hough_peaks<<<dim3(std::min(32, std::max(1, nr / peak_rows)), n_img, nc), std::min(256, (na + 63) / 64 * 64), 0, ctx->stream>>>(ctx->accum, ctx->peaks, ctx->counters, na, nr, threshold,
                                                                ctx->acc_cap, ctx->peak_cap, active, need_detect);
    KCHK("k_hough_peaks"); }
    if (K > 0) {
        Span sp(ctx, KID_TOPK, need_detect);
        k_hough_topk<<<dim3(n_img, nc), 256, 0, ctx->stream>>>(ctx->peaks, ctx->counters, ctx->lines, K, nr, (float)rho, (float)theta,
                                                               ctx->peak_cap, active, need_detect);
        KCHK("k_hough_topk");
    }
    return 0;
}

static int check_params(lfdmi_ctx *ctx, const lfdmi_params *p, bool dim) {
    if (!p) return fail(ctx, LFDMI_ERR_ARG, "params NULL");
    if (p->gaussKernel < 0 || p->gaussKernel > 31 || (p->gaussKernel > 0 && (p->gaussKernel & 1) == 0))
        return fail(ctx, LFDMI_ERR_ARG, "gaussKernel must be 0 (off) or odd, 1..31");
    if (p->nlinesInSet < 1 || p->nlinesInSet > LFDMI_MAX_SET_LINES) return fail(ctx, LFDMI_ERR_ARG, "nlinesInSet out of range");
    if (!p->dilateKernel) return fail(ctx, LFDMI_ERR_ARG, "dilateKernel NULL");
    if (dim && !p->erodeKernel) return fail(ctx, LFDMI_ERR_ARG, "erodeKernel NULL");
    return 0;
}

// Rows per band of k_prep_erode for this shape: as many as ctx->pe_rows while two 1024-thread workgroups fit a
// CU's LDS; 0 when the band would get so short that the halo rows dominate (wide frames with a tall erosion
// kernel, e.g. 4096 px x 9 rows: the separate kernels are the better choice there).
static int prep_erode_rows(const lfdmi_ctx *ctx, int w, int kh) {
    int BR = ctx->pe_rows;
    while (BR > 2 && (size_t)(2 * BR + kh - 1) * (w + 32) > 60 * 1024) BR >>= 1;
    if ((size_t)(2 * BR + kh - 1) * (w + 32) + 16 > 140 * 1024) return 0;
    return BR >= 2 * (kh - 1) ? BR : 0; // at most 50 % extra float reads
}

// prep + histogram + LUT + erosion without the 8-bit image in between (batch path of the dim pass); false if the
// shapes do not allow it
static bool can_fuse_prep_erode(const lfdmi_ctx *ctx, int dtype, int w, const uint8_t *kernel, int kh, int kw) {
    if (!ctx->fuse_prep_erode || ctx->keep_equ || dtype != LFDMI_F32 || (w % 16) != 0) return false; // (stage images wanted: keep gray)
    if (!kernel || kh <= 0 || kw <= 0 || kh > LFDMI_MAX_MORPH_K || kw > LFDMI_MAX_MORPH_K || !all_ones(kernel, kh, kw)) return false;
    return kw / 2 <= 16 && kw - 1 - kw / 2 <= 16 && prep_erode_rows(ctx, w, kh) > 0;
}

static int run_prep_erode(lfdmi_ctx *ctx, const void *src, int nc, int h, int w, int flip, int mode, double minFlux, double addFlux,
                          int kh, int kw, const int *active, bool from_bits = false) { // (hist / cellbm zeroed by the caller)
    int BR = prep_erode_rows(ctx, w, kh);
    size_t lds = (size_t)(2 * BR + kh - 1) * (w + 32) + 16; // (+ one piece: the sliding window peeks one word ahead)
    if (from_bits) { // the bright pass's image and bit planes instead of the float frames; histogram (hist2) taken there; marks into cellbm2
        Span sp(ctx, KID_BITS_ERODE);
        const int nwords = h * LFD_WQ(w);
        k_bits_erode<<<dim3((nwords + 256 * BE_REP - 1) / (256 * BE_REP), nc), 256, 0, ctx->stream>>>(ctx->gray, ctx->dbits, ctx->nzd, ctx->tmp, ctx->cellbm2,
                                                                             ctx->bm_bands, h, w, kh, kw, active);
        KCHK("k_bits_erode");
        return 0;
    }
    {
        Span sp(ctx, KID_PREP_ERODE);
#define LFD_PE_LAUNCH(M_)                                                                                                             \
    k_prep_erode<false, M_><<<dim3((h + BR - 1) / BR, nc), PE_THREADS, lds, ctx->stream>>>(                                           \
        (const float *)src, h, w, flip, mode, (float)minFlux, (float)addFlux, ctx->tmp, ctx->hist, kh, kw, BR, ctx->cellbm, ctx->bm_bands, \
        active, nullptr, nullptr, nullptr)
        switch (mode & 3) {
        case 0: LFD_PE_LAUNCH(0); break;
        case 1: LFD_PE_LAUNCH(1); break;
        case 2: LFD_PE_LAUNCH(2); break;
        default: LFD_PE_LAUNCH(3); break;
        }
#undef LFD_PE_LAUNCH
        KCHK("k_prep_erode");
    }
    Span sp(ctx, KID_LUT);
    k_lut<<<nc, 256, 0, ctx->stream>>>(ctx->hist, h * w, ctx->lut, active);
    KCHK("k_lut");
    return 0;
}

// Both passes' front ends in one sweep over the float frames (lfdmi_detect_batch): bright image + histogram + cell bitmap into
// gray / hist / cellbm, dim pass's eroded image + histogram + cell bitmap into tmp / hist2 / cellbm2; LUT of the bright pass.
static int run_prep_dual(lfdmi_ctx *ctx, const void *src, int nc, int h, int w, int flip, const lfdmi_params *dimp) {
    int kh = dimp->erode_kh, kw = dimp->erode_kw;
    int BR = prep_erode_rows(ctx, w, kh);
    size_t lds = (size_t)(2 * BR + kh - 1) * (w + 32) + 16;
    HIPCHK(hipMemsetAsync(ctx->zero_block2, 0, ctx->zero_bytes2, ctx->stream));
    {
        Span sp(ctx, KID_PREP_DUAL);
        k_prep_erode<true, LFDMI_PREP_BRIGHT_THEN_DIM><<<dim3((h + BR - 1) / BR, nc), PE_THREADS, lds, ctx->stream>>>((const float *)src, h, w, flip, LFDMI_PREP_BRIGHT_THEN_DIM,
                                                                              (float)dimp->minFlux, (float)dimp->addFlux, ctx->tmp, ctx->hist2, kh, kw, BR,
                                                                              ctx->cellbm2, ctx->bm_bands, nullptr, ctx->gray, ctx->hist, ctx->cellbm);
        KCHK("k_prep_dual");
    }
    Span sp(ctx, KID_LUT);
    k_lut<<<nc, 256, 0, ctx->stream>>>(ctx->hist, h * w, ctx->lut, nullptr);
    KCHK("k_lut");
    return 0;
}

// One detection pass on nc images already resident at src (device).  Front end: mask .. fit_minAreaRect (counters,
// bit rows, box image); tail: HoughLines on both images at one rho + check_theta -> res (device records of the slots).
static int run_front(lfdmi_ctx *ctx, const void *src, int dtype, int nc, int h, int w, int flip, int prep_mode, bool dim,
                     const lfdmi_params *p, const int *active, const lfdmi_params *dual_dim = nullptr) {
    const uint8_t *dil_src = ctx->gray;
    const u64 *bm = ctx->cellbm;
    ctx->eroded_valid = dim;
    ctx->perm_cur = nullptr;
    if (active && ctx->use_perm) { // the pass works on a subset of the slots: its active frames first (k_active_perm)
        k_active_perm<<<1, 64, 0, ctx->stream>>>(active, nc, ctx->perm);
        KCHK("k_active_perm");
        ctx->perm_cur = ctx->perm;
    }
    if (dim && ctx->delta_state == 2) {
        // the bright pass left this pass's values (gray + one bit per pixel) and their histogram: no second sweep over the floats
        HIPCHK(hipMemsetAsync(ctx->zero_block + ctx->zero_counters_off, 0, ctx->zero_bytes - ctx->zero_counters_off, ctx->stream));
        HIPCHK(hipMemsetAsync(ctx->cellbm2, 0, (size_t)nc * ctx->bm_bands * CELLBM_WORDS * sizeof(u64), ctx->stream));
        { Span sp(ctx, KID_LUT);
          k_lut<<<nc, 256, 0, ctx->stream>>>(ctx->hist2, h * w, ctx->lut, active);
          KCHK("k_lut"); }
        RET(run_prep_erode(ctx, nullptr, nc, h, w, 0, 0, 0, 0, p->erode_kh, p->erode_kw, active, true));
        dil_src = ctx->tmp;
        bm = ctx->cellbm2;
    } else
    if (dim && ctx->dual_state == 2) {
        // the bright pass's front end already produced this pass's eroded image, histogram and cell bitmap (run_prep_dual)
        HIPCHK(hipMemsetAsync(ctx->zero_block + ctx->zero_counters_off, 0, ctx->zero_bytes - ctx->zero_counters_off, ctx->stream));
        Span sp(ctx, KID_LUT);
        k_lut<<<nc, 256, 0, ctx->stream>>>(ctx->hist2, h * w, ctx->lut, active);
        KCHK("k_lut");
        dil_src = ctx->tmp;
        bm = ctx->cellbm2;
    } else {
    HIPCHK(hipMemsetAsync(ctx->zero_block, 0, ctx->zero_bytes, ctx->stream)); // counters, histograms, cell bitmap
    if (!dim && dual_dim && ctx->delta_state != 1) {
        RET(run_prep_dual(ctx, src, nc, h, w, flip, dual_dim));
    } else
    if (dim && can_fuse_prep_erode(ctx, dtype, w, p->erodeKernel, p->erode_kh, p->erode_kw)) {
        RET(run_prep_erode(ctx, src, nc, h, w, flip, prep_mode, p->minFlux, p->addFlux, p->erode_kh, p->erode_kw, active));
        dil_src = ctx->tmp;
    } else {
        // (dim: the prep kernel also marks the cells an erosion at least 7 wide can survive in, the erosion marks the cells of
        // its output: the fused dilate + Canny kernel then only visits tiles near what is left)
        const bool wide = dim && p->erode_kw >= 7 && dtype == LFDMI_F32 && (w & 3) == 0 && p->erode_kh <= 33;
        if (dim) HIPCHK(hipMemsetAsync(ctx->zero_block2, 0, ctx->zero_bytes2, ctx->stream));
        const lfdmi_params *dd = (!dim && ctx->delta_state == 1) ? dual_dim : nullptr;
        if (dd) HIPCHK(hipMemsetAsync(ctx->hist2, 0, (size_t)nc * 256 * sizeof(int), ctx->stream));
        RET(run_prep(ctx, src, dtype, nc, h, w, flip, prep_mode, p->minFlux, p->addFlux, active, true, wide ? ctx->fullbits : nullptr, dd));
        RET(rs_fill_point(ctx, 1));
        if (dim) {
            bool marked = false;
            // (the eroded plane is zero-filled only where the tile kernel can read it when that kernel is its one consumer)
            const bool only_tiles = wide && !ctx->keep_equ && ctx->sparse_erode_fill && p->gaussKernel <= 0 && ctx->use_cellbm && ctx->dc_tilelist &&
                                    p->dilate_kh <= 9 && p->dilate_kh <= DCW_MAXKH && can_fuse_dilate_canny(p->dilateKernel, p->dilate_kh, p->dilate_kw, w) &&
                                    (w + CANNY_TW - 1) / CANNY_TW <= 128 && (h + DCW_TH - 1) / DCW_TH <= DCT_MAXBANDS; // (the tile-list path)
            RET(run_morph(ctx, ctx->gray, ctx->tmp, nullptr, nullptr, p->erodeKernel, p->erode_kh, p->erode_kw, 1, nc, h, w, active,
                          wide ? ctx->fullbits : nullptr, ctx->cellbm2, &marked, only_tiles));
            if (only_tiles) ctx->eroded_valid = false; // (zero-filled only where the tile kernel reads it)
            dil_src = ctx->tmp;
            if (marked) bm = ctx->cellbm2;
        }
    }
    }
    if (p->gaussKernel > 0) { // optional smoothing of Canny's input (off in the reference): separate dilate, blur, Canny
        RET(run_morph(ctx, dil_src, ctx->equ, ctx->equb, ctx->lut, p->dilateKernel, p->dilate_kh, p->dilate_kw, 0, nc, h, w, active));
        RET(ensure_scratch(ctx, (size_t)ctx->G * ctx->N));
        RET(run_gauss(ctx, ctx->equ, (uint8_t *)ctx->scratch, nc, h, w, p->gaussKernel, p->gaussSigma, active));
        RET(run_canny(ctx, (const uint8_t *)ctx->scratch, nc, h, w, 0, 255, active, false, true));
    } else if (can_fuse_dilate_canny(p->dilateKernel, p->dilate_kh, p->dilate_kw, w)) {
        // the prep kernel of this pass left the cell occupancy of its output (a superset of the eroded image's)
        RET(run_dilate_canny(ctx, dil_src, nc, h, w, ctx->lut, p->dilate_kh, p->dilate_kw, active, ctx->use_cellbm ? bm : nullptr));
        RET(rs_fill_point(ctx, 2 + 10 * ctx->cur_pass));
        RET(run_canny(ctx, ctx->equ, nc, h, w, 0, 255, active, true, true));
    } else {
        RET(run_morph(ctx, dil_src, ctx->equ, ctx->equb, ctx->lut, p->dilateKernel, p->dilate_kh, p->dilate_kw, 0, nc, h, w, active));
        RET(run_canny(ctx, ctx->equ, nc, h, w, 0, 255, active, false, true));
    }
    return run_rects(ctx, nc, h, w, p->contoursMode, p->contoursMethod, p->minAreaRectMinLen, p->lwTresh, active);
}

static int run_tail(lfdmi_ctx *ctx, int nc, int h, int w, double rho, bool dim, const lfdmi_params *p, const int *active,
                    int *need_dim, lfdmi_result *res, bool again) {
    if (again) { // a further scale of the same front end
        k_hough_reset<<<(nc + 63) / 64, 64, 0, ctx->stream>>>(ctx->counters, nc);
        KCHK("k_hough_reset");
    }
    RET(run_hough(ctx, nc, h, w, rho, LFD_PI / 180, 1, 2, p->nlinesInSet, 1, active));
    TailParams tp;
    tp.navg = p->nlinesInSet; tp.dro = p->dro; tp.thetaTresh = p->thetaTresh; tp.lineSetTresh = p->lineSetTresh;
    tp.which = dim ? 2 : 1;
    Span sp(ctx, KID_FINALIZE);
    k_finalize<<<(nc + 63) / 64, 64, 0, ctx->stream>>>(ctx->lines, ctx->counters, res, need_dim, ctx->pass_flags, active, tp, nc);
    KCHK("k_finalize");
    return 0;
}

static int run_pass(lfdmi_ctx *ctx, const void *src, int dtype, int nc, int h, int w, int flip, int prep_mode, bool dim,
                    const lfdmi_params *p, const int *active, int *need_dim, const lfdmi_params *dual_dim = nullptr) {
    RET(run_front(ctx, src, dtype, nc, h, w, flip, prep_mode, dim, p, active, dual_dim));
    ctx->perm_cur = nullptr; // (only the front end's per-frame launches use it)
    return run_tail(ctx, nc, h, w, p->houghMethod, dim, p, active, need_dim, ctx->res_dev, false);
}

static void dictify(int h, int w, lfdmi_result *r) { // processfield.py:266-288, float32 scalars
    if (!r->found) return;
    float c = cosf(r->theta), s = sinf(r->theta);
    float x0 = c * r->rho, y0 = s * r->rho;
    float L = (float)(h + w);
    r->x1 = (int32_t)(x0 - L * s);
    r->y1 = (int32_t)(y0 + L * c);
    r->x2 = (int32_t)(x0 + L * s);
    r->y2 = (int32_t)(y0 - L * c);
}

// ---- C-ABI: per-operator entry points -------------------------------------------------------
extern "C" int lfdmi_prep_u8(lfdmi_ctx *ctx, const void *src, int dtype, int n, int h, int w, int flip, int mode,
                             double minFlux, double addFlux, uint8_t *gray, int32_t *hist, int loc) {
    RET(check_shape(ctx, n, h, w));
    if (!src || dtype < 0 || dtype > 2 || mode < 0 || mode > 3) return fail(ctx, LFDMI_ERR_ARG, "lfdmi_prep_u8: bad argument");
    if (dtype == LFDMI_U8 && (mode & 2)) return fail(ctx, LFDMI_ERR_DTYPE, "dim masking needs a float image (numpy refuses uint8 += float)");
    size_t N = (size_t)h * w, es = dtype_size(dtype);
    for (int c0 = 0; c0 < n; c0 += ctx->G) {
        int nc = n - c0 < ctx->G ? n - c0 : ctx->G;
        const void *d;
        RET(in_ptr(ctx, src, (size_t)c0 * N * es, (size_t)nc * N * es, loc, &d));
        RET(run_prep(ctx, d, dtype, nc, h, w, flip, mode, minFlux, addFlux, nullptr));
        RET(out_copy(ctx, gray, (size_t)c0 * N, ctx->gray, (size_t)nc * N, loc));
        RET(out_copy(ctx, hist, (size_t)c0 * 256 * 4, ctx->hist, (size_t)nc * 256 * 4, loc));
        RET(sync(ctx, nc));
    }
    return 0;
}

extern "C" int lfdmi_equalize_hist(lfdmi_ctx *ctx, const uint8_t *src, int n, int h, int w, uint8_t *dst, int loc) {
    RET(check_shape(ctx, n, h, w));
    if (!src || !dst) return fail(ctx, LFDMI_ERR_ARG, "NULL image");
    size_t N = (size_t)h * w;
    for (int c0 = 0; c0 < n; c0 += ctx->G) {
        int nc = n - c0 < ctx->G ? n - c0 : ctx->G;
        const void *d;
        RET(in_ptr(ctx, src, (size_t)c0 * N, (size_t)nc * N, loc, &d));
        {
            Span sp(ctx, KID_MISC);
            HIPCHK(hipMemsetAsync(ctx->hist, 0, (size_t)nc * 256 * sizeof(int), ctx->stream));
            k_hist_u8<<<dim3(256, nc), 256, 0, ctx->stream>>>((const uint8_t *)d, N, ctx->hist);
            KCHK("k_hist_u8");
            k_lut<<<nc, 256, 0, ctx->stream>>>(ctx->hist, h * w, ctx->lut, nullptr);
            KCHK("k_lut");
            k_apply_lut<<<dim3(512, nc), 256, 0, ctx->stream>>>((const uint8_t *)d, ctx->lut, ctx->equ, N);
            KCHK("k_apply_lut");
        }
        RET(out_copy(ctx, dst, (size_t)c0 * N, ctx->equ, (size_t)nc * N, loc));
        RET(sync(ctx, nc));
    }
    return 0;
}

static int morph_api(lfdmi_ctx *ctx, const uint8_t *src, int n, int h, int w, const uint8_t *kernel, int kh, int kw,
                     uint8_t *dst, int loc, int op) {
    RET(check_shape(ctx, n, h, w));
    if (!src || !dst) return fail(ctx, LFDMI_ERR_ARG, "NULL image");
    size_t N = (size_t)h * w;
    for (int c0 = 0; c0 < n; c0 += ctx->G) {
        int nc = n - c0 < ctx->G ? n - c0 : ctx->G;
        const void *d;
        RET(in_ptr(ctx, src, (size_t)c0 * N, (size_t)nc * N, loc, &d));
        RET(run_morph(ctx, (const uint8_t *)d, ctx->equ, nullptr, nullptr, kernel, kh, kw, op, nc, h, w, nullptr));
        RET(out_copy(ctx, dst, (size_t)c0 * N, ctx->equ, (size_t)nc * N, loc));
        RET(sync(ctx, nc));
    }
    return 0;
}
extern "C" int lfdmi_dilate(lfdmi_ctx *ctx, const uint8_t *src, int n, int h, int w, const uint8_t *kernel, int kh, int kw,
                            uint8_t *dst, int loc) { return morph_api(ctx, src, n, h, w, kernel, kh, kw, dst, loc, 0); }
extern "C" int lfdmi_erode(lfdmi_ctx *ctx, const uint8_t *src, int n, int h, int w, const uint8_t *kernel, int kh, int kw,
                           uint8_t *dst, int loc) { return morph_api(ctx, src, n, h, w, kernel, kh, kw, dst, loc, 1); }

extern "C" int lfdmi_gaussian_blur(lfdmi_ctx *ctx, const uint8_t *src, int n, int h, int w, int ksize, double sigma, uint8_t *dst, int loc) {
    RET(check_shape(ctx, n, h, w));
    if (!src || !dst) return fail(ctx, LFDMI_ERR_ARG, "NULL image");
    size_t N = (size_t)h * w;
    for (int c0 = 0; c0 < n; c0 += ctx->G) {
        int nc = n - c0 < ctx->G ? n - c0 : ctx->G;
        const void *d;
        RET(in_ptr(ctx, src, (size_t)c0 * N, (size_t)nc * N, loc, &d));
        RET(run_gauss(ctx, (const uint8_t *)d, ctx->equ, nc, h, w, ksize, sigma, nullptr));
        RET(out_copy(ctx, dst, (size_t)c0 * N, ctx->equ, (size_t)nc * N, loc));
        RET(sync(ctx, nc));
    }
    return 0;
}

static int expand_bits(lfdmi_ctx *ctx, const u64 *bits, uint8_t *dev_dst, int nc, int h, int w) {
    k_u8_from_bits<<<dim3((w + 63) / 64, (h + 3) / 4, nc), 256, 0, ctx->stream>>>(bits, dev_dst, h, w);
    KCHK("k_u8_from_bits");
    return 0;
}

extern "C" int lfdmi_canny(lfdmi_ctx *ctx, const uint8_t *src, int n, int h, int w, double low, double high, uint8_t *dst, int loc) {
    RET(check_shape(ctx, n, h, w));
    if (!src || !dst) return fail(ctx, LFDMI_ERR_ARG, "NULL image");
    size_t N = (size_t)h * w;
    std::vector<int> cnt((size_t)ctx->G * C_COUNT), pflags((size_t)ctx->G);
    for (int c0 = 0; c0 < n; c0 += ctx->G) {
        int nc = n - c0 < ctx->G ? n - c0 : ctx->G;
        const void *d;
        RET(in_ptr(ctx, src, (size_t)c0 * N, (size_t)nc * N, loc, &d));
        RET(zero_counters(ctx, nc));
        HIPCHK(hipMemsetAsync(ctx->pass_flags, 0, (size_t)nc * sizeof(int), ctx->stream));
        RET(run_canny(ctx, (const uint8_t *)d, nc, h, w, low, high, nullptr));
        RET(expand_bits(ctx, ctx->edgeb, ctx->tmp, nc, h, w));
        RET(out_copy(ctx, dst, (size_t)c0 * N, ctx->tmp, (size_t)nc * N, loc));
        HIPCHK(hipMemcpyAsync(cnt.data(), ctx->counters, (size_t)nc * C_COUNT * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipMemcpyAsync(pflags.data(), ctx->pass_flags, (size_t)nc * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        RET(sync(ctx, nc));
        if (scan_gave_up(ctx, pflags.data(), nc)) { c0 -= ctx->G; continue; } // the look-back scan gave up: this chunk again, three launches per scan
        chunk_done(ctx);
        for (int i = 0; i < nc; i++)
            if (cnt[(size_t)i * C_COUNT + C_OVERFLOW]) { // more runs than this workspace's tables hold: the worst-case one takes the image
                lfdmi_ctx *sp = get_spill(ctx);
                if (!sp) return fail(ctx, LFDMI_ERR_CAPACITY, "run tables too small for this image");
                int rc = lfdmi_canny(sp, src + (size_t)(c0 + i) * N, 1, h, w, low, high, dst + (size_t)(c0 + i) * N, loc);
                if (rc) { ctx->err = sp->err; return rc; }
                ctx->n_spilled++;
            }
    }
    return 0;
}

extern "C" int lfdmi_fit_min_area_rect(lfdmi_ctx *ctx, const uint8_t *img, int n, int h, int w, int contoursMode,
                                       int contoursMethod, double minAreaRectMinLen, double lwTresh, uint8_t *box_img,
                                       int32_t *detection, int32_t *n_boxes, int loc) {
    RET(check_shape(ctx, n, h, w));
    if (!img) return fail(ctx, LFDMI_ERR_ARG, "NULL image");
    size_t N = (size_t)h * w;
    std::vector<int> cnt((size_t)ctx->G * C_COUNT), pflags((size_t)ctx->G);
    for (int c0 = 0; c0 < n; c0 += ctx->G) {
        int nc = n - c0 < ctx->G ? n - c0 : ctx->G;
        const void *d;
        RET(in_ptr(ctx, img, (size_t)c0 * N, (size_t)nc * N, loc, &d));
        RET(zero_counters(ctx, nc));
        HIPCHK(hipMemsetAsync(ctx->pass_flags, 0, (size_t)nc * sizeof(int), ctx->stream));
        RET(run_canny(ctx, (const uint8_t *)d, nc, h, w, 0, 255, nullptr, false, true));
        RET(run_rects(ctx, nc, h, w, contoursMode, contoursMethod, minAreaRectMinLen, lwTresh, nullptr));
        if (box_img) {
            RET(expand_bits(ctx, ctx->boxb, ctx->tmp, nc, h, w));
            RET(out_copy(ctx, box_img, (size_t)c0 * N, ctx->tmp, (size_t)nc * N, loc));
        }
        HIPCHK(hipMemcpyAsync(cnt.data(), ctx->counters, (size_t)nc * C_COUNT * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipMemcpyAsync(pflags.data(), ctx->pass_flags, (size_t)nc * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        RET(sync(ctx, nc));
        if (scan_gave_up(ctx, pflags.data(), nc)) { c0 -= ctx->G; continue; } // (see lfdmi_canny)
        chunk_done(ctx);
        for (int i = 0; i < nc; i++) {
            if (cnt[(size_t)i * C_COUNT + C_OVERFLOW]) { // a table of this workspace is too small for the image: the worst-case one takes it
                lfdmi_ctx *sp = get_spill(ctx);
                if (!sp) return fail(ctx, LFDMI_ERR_CAPACITY, "contour workspace overflow");
                int rc = lfdmi_fit_min_area_rect(sp, img + (size_t)(c0 + i) * N, 1, h, w, contoursMode, contoursMethod, minAreaRectMinLen, lwTresh,
                                                 box_img ? box_img + (size_t)(c0 + i) * N : nullptr, detection ? detection + c0 + i : nullptr,
                                                 n_boxes ? n_boxes + c0 + i : nullptr, loc);
                if (rc) { ctx->err = sp->err; return rc; }
                ctx->n_spilled++;
                continue;
            }
            int det = cnt[(size_t)i * C_COUNT + C_DETECT], nb = cnt[(size_t)i * C_COUNT + C_NQUADS];
            if (loc == LFDMI_DEVICE) {
                if (detection) HIPCHK(hipMemcpy(detection + c0 + i, &det, 4, hipMemcpyHostToDevice));
                if (n_boxes) HIPCHK(hipMemcpy(n_boxes + c0 + i, &nb, 4, hipMemcpyHostToDevice));
            } else {
                if (detection) detection[c0 + i] = det;
                if (n_boxes) n_boxes[c0 + i] = nb;
            }
        }
    }
    return 0;
}

static int hough_api(lfdmi_ctx *ctx, const uint8_t *img, int n, int h, int w, double rho, double theta, int threshold,
                     int max_lines, float *lines, int32_t *n_lines, int32_t *accum, int loc) {
    RET(check_shape(ctx, n, h, w));
    if (!img) return fail(ctx, LFDMI_ERR_ARG, "NULL image");
    if (!hough_fits(ctx, h, w, rho, theta) && !ctx->worst) { // finer than this workspace's accumulators: the worst-case one takes the call
        lfdmi_ctx *sp = get_spill(ctx);
        if (!sp) return fail(ctx, LFDMI_ERR_CAPACITY, "Hough accumulator larger than the workspace");
        int rc = hough_api(sp, img, n, h, w, rho, theta, threshold, max_lines, lines, n_lines, accum, loc);
        if (rc) ctx->err = sp->err;
        return rc;
    }
    RET(ensure_tables(ctx, h, w, rho, theta));
    size_t N = (size_t)h * w;
    int na = ctx->numangle, nr = ctx->numrho;
    size_t acc_n = (size_t)(na + 2) * (nr + 2);
    std::vector<int> cnt((size_t)ctx->G * C_COUNT);
    float *lines_dev = nullptr;
    size_t lines_bytes = (lines && max_lines > 0) ? (size_t)ctx->G * max_lines * 2 * sizeof(float) : 0;
    size_t acc_bytes = accum ? (size_t)ctx->G * acc_n * sizeof(int) : 0;
    if (lines_bytes + acc_bytes) RET(ensure_scratch(ctx, lines_bytes + acc_bytes)); // sorted lines | untransposed accumulators
    if (lines_bytes) lines_dev = (float *)ctx->scratch;
    for (int c0 = 0; c0 < n; c0 += ctx->G) {
        int nc = n - c0 < ctx->G ? n - c0 : ctx->G;
        const void *d;
        RET(in_ptr(ctx, img, (size_t)c0 * N, (size_t)nc * N, loc, &d));
        RET(zero_counters(ctx, nc));
        k_bits_from_u8<<<dim3((w + 63) / 64, (h + 3) / 4, nc), 256, 0, ctx->stream>>>((const uint8_t *)d, ctx->equb, h, w);
        KCHK("k_bits_from_u8");
        RET(run_hough(ctx, nc, h, w, rho, theta, threshold, 1, 0, 0, nullptr));
        if (accum) {
            int *tmp_acc = (int *)((char *)ctx->scratch + lines_bytes);
            k_accum_untranspose<<<dim3(64, nc), 256, 0, ctx->stream>>>(ctx->accum, tmp_acc, na, nr, ctx->acc_cap);
            KCHK("k_accum_untranspose");
            RET(out_copy(ctx, accum, (size_t)c0 * acc_n * 4, tmp_acc, (size_t)nc * acc_n * 4, loc));
        }
        if (lines_dev) {
            Span sp(ctx, KID_SORT);
            k_hough_sort<<<dim3(1, nc), 1024, 0, ctx->stream>>>(ctx->peaks, ctx->counters, lines_dev, max_lines, nr, (float)rho,
                                                               (float)theta, ctx->peak_cap);
            KCHK("k_hough_sort");
        }
        HIPCHK(hipMemcpyAsync(cnt.data(), ctx->counters, (size_t)nc * C_COUNT * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        RET(sync(ctx, nc));
        for (int i = 0; i < nc; i++) {
            if (cnt[(size_t)i * C_COUNT + C_OVERFLOW]) { // chunk list or peak list too small for this image: the worst-case workspace takes it
                lfdmi_ctx *sp = get_spill(ctx);
                if (!sp) return fail(ctx, LFDMI_ERR_CAPACITY, "Hough lists too small for this image");
                int rc = hough_api(sp, img + (size_t)(c0 + i) * N, 1, h, w, rho, theta, threshold, max_lines,
                                   lines ? lines + (size_t)(c0 + i) * max_lines * 2 : nullptr, n_lines ? n_lines + c0 + i : nullptr,
                                   accum ? accum + (size_t)(c0 + i) * acc_n : nullptr, loc);
                if (rc) { ctx->err = sp->err; return rc; }
                ctx->n_spilled++;
                continue;
            }
            int total = cnt[(size_t)i * C_COUNT + C_NPEAK_EQU];
            if (n_lines) {
                if (loc == LFDMI_DEVICE) HIPCHK(hipMemcpy(n_lines + c0 + i, &total, 4, hipMemcpyHostToDevice));
                else n_lines[c0 + i] = total;
            }
            if (lines_dev) {
                int m = total < max_lines ? total : max_lines;
                if (m > 0)
                    HIPCHK(hipMemcpy((char *)lines + (size_t)(c0 + i) * max_lines * 8, lines_dev + (size_t)i * max_lines * 2, (size_t)m * 8,
                                     loc == LFDMI_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost));
            }
        }
    }
    return 0;
}

extern "C" int lfdmi_hough_lines(lfdmi_ctx *ctx, const uint8_t *img, int n, int h, int w, double rho, double theta, int threshold,
                                 int max_lines, float *lines, int32_t *n_lines, int loc) {
    return hough_api(ctx, img, n, h, w, rho, theta, threshold, max_lines, lines, n_lines, nullptr, loc);
}
extern "C" int lfdmi_hough_accum(lfdmi_ctx *ctx, const uint8_t *img, int n, int h, int w, double rho, double theta, int32_t *accum, int loc) {
    return hough_api(ctx, img, n, h, w, rho, theta, 1, 0, nullptr, nullptr, accum, loc);
}

// catalogue arrays -> device; returns device pointers in *dev (struct copy)
static int stage_catalog(lfdmi_ctx *ctx, const lfdmi_catalog *cat, int f0, int nc, lfdmi_catalog *dev) {
    size_t m = (size_t)cat->max_obj;
    *dev = *cat;
    if (cat->loc == LFDMI_DEVICE) {
        dev->count = cat->count + f0;
        dev->rowc = cat->rowc + f0 * m * 5; dev->colc = cat->colc + f0 * m * 5;
        dev->psfmag = cat->psfmag + f0 * m * 5; dev->petro90 = cat->petro90 + f0 * m * 5;
        dev->nobserve = cat->nobserve + f0 * m; dev->ndetect = cat->ndetect + f0 * m;
        return 0;
    }
    size_t b5 = (size_t)nc * m * 5 * 4, b1 = (size_t)nc * m * 4, bc = (size_t)nc * 4;
    size_t total = 4 * b5 + 2 * b1 + bc + 256;
    if (ctx->cat_bytes < total) {
        if (ctx->cat_dev) { HIPCHK(hipStreamSynchronize(ctx->stream)); HIPCHK(hipFree(ctx->cat_dev)); ctx->cat_dev = nullptr; ctx->cat_bytes = 0; }
        HIPCHK(hipMalloc(&ctx->cat_dev, total));
        ctx->cat_bytes = total;
    }
    char *p = (char *)ctx->cat_dev;
    auto up = [&](const void *src, size_t off_elems_bytes, size_t bytes, const void **out) -> int {
        HIPCHK(hipMemcpyAsync(p, (const char *)src + off_elems_bytes, bytes, hipMemcpyHostToDevice, ctx->stream));
        *out = p;
        p += (bytes + 15) & ~(size_t)15;
        return 0;
    };
    RET(up(cat->count, (size_t)f0 * 4, bc, (const void **)&dev->count));
    RET(up(cat->rowc, f0 * m * 20, b5, (const void **)&dev->rowc));
    RET(up(cat->colc, f0 * m * 20, b5, (const void **)&dev->colc));
    RET(up(cat->psfmag, f0 * m * 20, b5, (const void **)&dev->psfmag));
    RET(up(cat->petro90, f0 * m * 20, b5, (const void **)&dev->petro90));
    RET(up(cat->nobserve, f0 * m * 4, b1, (const void **)&dev->nobserve));
    RET(up(cat->ndetect, f0 * m * 4, b1, (const void **)&dev->ndetect));
    return 0;
}

static int run_removestars(lfdmi_ctx *ctx, float *frames_dev, int f0, int nc, int h, int w, const lfdmi_catalog *cat,
                           const lfdmi_rs_params *rs, std::vector<int4> *host_boxes = nullptr, bool fill = true) {
    if (!cat || cat->max_obj <= 0) return 0;
    if (!rs || rs->filter_index < 0 || rs->filter_index > 4) return fail(ctx, LFDMI_ERR_ARG, "removestars params");
    lfdmi_catalog dev;
    RET(stage_catalog(ctx, cat, f0, nc, &dev));
    RsDev p;
    p.defaultxy = rs->defaultxy; p.maxxy = rs->maxxy; p.magcount = rs->magcount; p.filter_index = rs->filter_index;
    p.pixscale = rs->pixscale; p.maxmagdiff = rs->maxmagdiff; p.filter_cap = rs->filter_cap;
    size_t need = (size_t)nc * cat->max_obj;
    if (ctx->rs_boxes_cap < need) {
        if (ctx->rs_boxes) { HIPCHK(hipStreamSynchronize(ctx->stream)); HIPCHK(hipFree(ctx->rs_boxes)); ctx->rs_boxes = nullptr; ctx->rs_boxes_cap = 0; }
        if (ctx->rs_sboxes) { HIPCHK(hipFree(ctx->rs_sboxes)); ctx->rs_sboxes = nullptr; }
        HIPCHK(hipMalloc(&ctx->rs_boxes, need * sizeof(int4)));
        ctx->rs_boxes_cap = need;
    }
    // crowded catalogues: squares sorted by first row, so that the sweep's fold and the zero fill only look at the squares near
    // their rows and every pixel is blotted once (k_image.h: k_rs_sort)
    ctx->rs_sorted = cat->max_obj > ctx->rs_sort_min && h < RS_SORT_MAXH && (w % 32) == 0 && w <= RS_MAXW;
    if (ctx->rs_sorted) {
        if (!ctx->rs_sboxes) HIPCHK(hipMalloc(&ctx->rs_sboxes, ctx->rs_boxes_cap * sizeof(int4)));
        const size_t nrs = (size_t)nc * (h + 1);
        if (ctx->rs_rowstart_cap < nrs) {
            if (ctx->rs_rowstart) { HIPCHK(hipStreamSynchronize(ctx->stream)); HIPCHK(hipFree(ctx->rs_rowstart)); ctx->rs_rowstart = nullptr; }
            HIPCHK(hipMalloc(&ctx->rs_rowstart, nrs * sizeof(int)));
            ctx->rs_rowstart_cap = nrs;
        }
        ctx->rs_hmax = 2 * std::max(rs->maxxy, rs->defaultxy); // a square's side: 2 dxy, dxy <= maxxy or the default
    }
    int4 *boxes = ctx->rs_boxes;
    if (host_boxes) host_boxes->resize(need); // the blotted squares come back instead of the blotted frames
    {
        Span sp(ctx, KID_REMOVESTARS);
        k_rs_boxes<<<dim3((cat->max_obj + 255) / 256, nc), 256, 0, ctx->stream>>>(h, w, cat->max_obj, dev.count, dev.rowc, dev.colc, dev.psfmag,
                                                                               dev.petro90, dev.nobserve, dev.ndetect, p, boxes);
        KCHK("k_rs_boxes");
        ctx->rs_count_dev = dev.count;
        ctx->rs_max_obj = cat->max_obj;
        if (ctx->rs_sorted) {
            k_rs_sort<<<nc, 1024, 0, ctx->stream>>>(h, cat->max_obj, dev.count, boxes, ctx->rs_sboxes, ctx->rs_rowstart);
            KCHK("k_rs_sort");
        }
        if (fill) {
            if (ctx->rs_sorted)
                k_rs_fill_bands<<<dim3((h + RS_BAND_ROWS - 1) / RS_BAND_ROWS, nc), 256, 0, ctx->stream>>>(frames_dev, h, w, cat->max_obj, ctx->rs_sboxes,
                                                                                                     ctx->rs_rowstart, ctx->rs_hmax);
            else
                k_rs_fill<<<dim3((cat->max_obj + 3) / 4, nc), 256, 0, ctx->stream>>>(frames_dev, h, w, cat->max_obj, dev.count, boxes, 0);
            KCHK("k_rs_fill");
        }
    }
    if (host_boxes)
        HIPCHK(hipMemcpyAsync(host_boxes->data(), boxes, host_boxes->size() * sizeof(int4), hipMemcpyDeviceToHost, ctx->stream));
    return 0;
}

// remove_stars on the caller's HOST frames: the squares the device computed (and zero-filled in its copy) are
// zero-filled in the caller's array by host threads, while the GPU runs the rest of the pipe -- the blotted
// frames are not copied back (24.4 MB over PCIe per frame instead of 12.2 MB halves the host-fed rate)
static void bind_thread(const std::vector<int> &cpus);
static void blot_host_frames(float *frames, int nc, int h, int w, const lfdmi_catalog *cat, int f0, const std::vector<int4> &boxes,
                             const std::vector<int> &cpus) {
    size_t N = (size_t)h * w;
    auto work = [&](int a, int b) {
        bind_thread(cpus);
        for (int f = a; f < b; f++) {
            int n_obj = cat->count[f0 + f];
            float *img = frames + (size_t)f * N;
            for (int i = 0; i < n_obj && i < cat->max_obj; i++) {
                int4 bx = boxes[(size_t)f * cat->max_obj + i];
                if (bx.y <= bx.x || bx.w <= bx.z) continue;
                for (int r = bx.x; r < bx.y; r++) memset(img + (size_t)r * w + bx.z, 0, (size_t)(bx.w - bx.z) * sizeof(float));
            }
        }
    };
    int nt = nc >= 32 ? 8 : (nc >= 8 ? 4 : 1);
    if (nt == 1) { work(0, nc); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < nt; t++) th.emplace_back(work, nc * t / nt, nc * (t + 1) / nt);
    for (auto &x : th) x.join();
}

extern "C" int lfdmi_remove_stars(lfdmi_ctx *ctx, float *img, int n, int h, int w, const lfdmi_catalog *cat,
                                  const lfdmi_rs_params *rs, int loc) {
    RET(check_shape(ctx, n, h, w));
    if (!img || !cat) return fail(ctx, LFDMI_ERR_ARG, "NULL argument");
    size_t N = (size_t)h * w;
    for (int c0 = 0; c0 < n; c0 += ctx->G) {
        int nc = n - c0 < ctx->G ? n - c0 : ctx->G;
        const void *d;
        RET(in_ptr(ctx, img, (size_t)c0 * N * 4, (size_t)nc * N * 4, loc, &d));
        RET(run_removestars(ctx, (float *)d, c0, nc, h, w, cat, rs));
        if (loc == LFDMI_HOST) RET(out_copy(ctx, img, (size_t)c0 * N * 4, d, (size_t)nc * N * 4, loc));
        RET(sync(ctx, nc));
    }
    return 0;
}

// ---- C-ABI: whole passes ------------------------------------------------------------------
// Whole-pass entry points launch the general run kernels only once the context has met a frame the per-frame
// LDS kernels could not take; a chunk that meets the first such frame is run again with them.
#define GENERAL_QUIET_CHUNKS 8
struct GeneralGuard {
    lfdmi_ctx *c;
    explicit GeneralGuard(lfdmi_ctx *ctx) : c(ctx) { c->general_on = !c->frame_ccl || c->general_seen; }
    ~GeneralGuard() { c->general_on = true; }
    bool again(const int *flags, int n) {
        bool need = false;
        for (int i = 0; i < n; i++) need = need || (flags[i] & PASS_FLAG_GENERAL);
        if (c->general_on) {
            c->n_general_chunks++;
            // the ~22 extra launches per pass are dropped again once GENERAL_QUIET_CHUNKS chunks in a row did without them
            if (need) c->quiet_chunks = 0;
            else if (c->frame_ccl && ++c->quiet_chunks >= GENERAL_QUIET_CHUNKS) { c->general_seen = false; c->quiet_chunks = 0; }
            return false;
        }
        if (!need) return false;
        c->general_seen = true;
        c->general_on = true;
        c->quiet_chunks = 0;
        c->n_general_reruns++;
        return true;
    }
};

// results[s * rstride + i]: frame i at Hough scale s (rhos == nullptr: one scale, p->houghMethod)
struct KeepEqu { lfdmi_ctx *c; bool old; KeepEqu(lfdmi_ctx *c_, bool v) : c(c_), old(c_->keep_equ) { c->keep_equ = v; } ~KeepEqu() { c->keep_equ = old; } };

static int pass_api(lfdmi_ctx *ctx, const void *img, int dtype, int n, int h, int w, int flip, int prep_mode, bool dim,
                    const lfdmi_params *p, int n_scales, const double *rhos, lfdmi_result *results, size_t rstride, float *lines_equ,
                    float *lines_box, int loc) {
    RET(check_shape(ctx, n, h, w));
    RET(check_params(ctx, p, dim));
    if (!img || !results || dtype < 0 || dtype > 2) return fail(ctx, LFDMI_ERR_ARG, "bad argument");
    if (n_scales < 1 || n_scales > LFDMI_MAX_SCALES) return fail(ctx, LFDMI_ERR_ARG, "n_scales out of range");
    if (dtype == LFDMI_U8 && dim) return fail(ctx, LFDMI_ERR_DTYPE, "dim pass needs a float image (numpy refuses uint8 += float)");
    const double rho1 = p->houghMethod;
    if (!rhos) rhos = &rho1;
    for (int s = 0; s < n_scales; s++)
        if (!hough_fits(ctx, h, w, rhos[s], LFD_PI / 180)) { // finer than this workspace's accumulators: the worst-case one takes the call
            lfdmi_ctx *sp = get_spill(ctx);
            if (!sp || !hough_fits(sp, h, w, rhos[s], LFD_PI / 180)) return fail(ctx, LFDMI_ERR_CAPACITY, "Hough accumulator larger than the workspace (rho < 1 px)");
            int rc = pass_api(sp, img, dtype, n, h, w, flip, prep_mode, dim, p, n_scales, rhos, results, rstride, lines_equ, lines_box, loc);
            if (rc) ctx->err = sp->err;
            return rc;
        }
    size_t N = (size_t)h * w, es = dtype_size(dtype);
    int K = p->nlinesInSet;
    const size_t G = (size_t)ctx->G;
    KeepEqu keep_guard(ctx, ctx->stage_mode != 0); // the 8-bit stage images for lfdmi_get_stage unless switched off (batches)
    ctx->stages_valid = ctx->keep_equ;
    std::vector<lfdmi_result> host(G * n_scales);
    std::vector<float> hl(G * 2 * K * 2);
    std::vector<int> flags(G);
    ctx->cur_pass = 0;
    for (int c0 = 0; c0 < n; c0 += ctx->G) {
        int nc = n - c0 < ctx->G ? n - c0 : ctx->G;
        const void *d;
        RET(in_ptr(ctx, img, (size_t)c0 * N * es, (size_t)nc * N * es, loc, &d));
        GeneralGuard gg(ctx);
        int grow_tries = 0;
        for (;;) { // (again, with the general run kernels, if a frame turned out to need them)
            for (int s = 0; s < n_scales; s++) {
                k_init_results<<<(nc + 63) / 64, 64, 0, ctx->stream>>>(ctx->res_dev + s * G, ctx->pass_flags, nc);
                KCHK("k_init_results");
            }
            RET(run_front(ctx, d, dtype, nc, h, w, flip, prep_mode, dim, p, nullptr));
            for (int s = 0; s < n_scales; s++) {
                RET(run_tail(ctx, nc, h, w, rhos[s], dim, p, nullptr, nullptr, ctx->res_dev + s * G, s > 0));
                HIPCHK(hipMemcpyAsync(host.data() + s * G, ctx->res_dev + s * G, (size_t)nc * sizeof(lfdmi_result), hipMemcpyDeviceToHost, ctx->stream));
            }
            HIPCHK(hipMemcpyAsync(hl.data(), ctx->lines, (size_t)nc * 2 * K * 2 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(hipMemcpyAsync(flags.data(), ctx->pass_flags, (size_t)nc * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(hipStreamSynchronize(ctx->stream));
            if (scan_gave_up(ctx, flags.data(), nc)) continue; // (see lfdmi_detect_batch)
            if (grow_tries < 4 && grow_caps(ctx, host.data(), nc, G, n_scales)) { grow_tries++; continue; } // tables enlarged: once more
            if (!gg.again(flags.data(), nc)) break;
        }
        chunk_done(ctx);
        {
            int na[2] = {nc, 0}, nd[2] = {0, 0};
            for (int i = 0; i < nc; i++) nd[0] += (flags[i] & (dim ? 4 : 1)) != 0;
            RET(sync(ctx, nc, na, nd));
        }
        for (int i = 0; i < nc; i++) {
            bool spill = false;
            for (int s = 0; s < n_scales; s++) spill = spill || host[s * G + i].status == LFDMI_ERR_CAPACITY;
            if (spill) { // a table of this workspace was too small for the frame: once more, alone, in the worst-case workspace
                lfdmi_ctx *sp = get_spill(ctx);
                if (sp) {
                    lfdmi_result one[LFDMI_MAX_SCALES];
                    int rc = pass_api(sp, (const char *)img + (size_t)(c0 + i) * N * es, dtype, 1, h, w, flip, prep_mode, dim, p, n_scales, rhos,
                                      one, 1, lines_equ ? lines_equ + (size_t)(c0 + i) * 2 * K : nullptr,
                                      lines_box ? lines_box + (size_t)(c0 + i) * 2 * K : nullptr, loc);
                    if (rc) { ctx->err = sp->err; return rc; }
                    for (int s = 0; s < n_scales; s++) results[s * rstride + c0 + i] = one[s];
                    ctx->n_spilled++;
                    continue;
                }
            }
            for (int s = 0; s < n_scales; s++) {
                dictify(h, w, &host[s * G + i]);
                results[s * rstride + c0 + i] = host[s * G + i];
            }
            const lfdmi_result &last = host[(size_t)(n_scales - 1) * G + i]; // (the lines of the last scale are the ones still in the workspace)
            bool have = last.detection && last.status == 0;
            for (int s = 0; s < 2; s++) {
                float *dst = s ? lines_box : lines_equ;
                if (!dst) continue;
                for (int k = 0; k < 2 * K; k++) dst[(size_t)(c0 + i) * 2 * K + k] = have ? hl[((size_t)i * 2 + s) * 2 * K + k] : 0.f;
            }
        }
    }
    return 0;
}

extern "C" int lfdmi_process_bright(lfdmi_ctx *ctx, const void *img, int dtype, int n, int h, int w, int flip,
                                    const lfdmi_params *p, lfdmi_result *results, float *lines_equ, float *lines_box, int loc) {
    return pass_api(ctx, img, dtype, n, h, w, flip, LFDMI_PREP_BRIGHT, false, p, 1, nullptr, results, (size_t)n, lines_equ, lines_box, loc);
}

extern "C" int lfdmi_process_dim(lfdmi_ctx *ctx, const void *img, int dtype, int n, int h, int w, int flip, int after_bright,
                                 const lfdmi_params *p, lfdmi_result *results, float *lines_equ, float *lines_box, int loc) {
    return pass_api(ctx, img, dtype, n, h, w, flip, after_bright ? LFDMI_PREP_BRIGHT_THEN_DIM : LFDMI_PREP_DIM, true, p, 1, nullptr, results,
                    (size_t)n, lines_equ, lines_box, loc);
}

extern "C" int lfdmi_process_multiscale(lfdmi_ctx *ctx, const void *img, int dtype, int n, int h, int w, int flip, int dim, int after_bright,
                                        const lfdmi_params *p, int n_scales, const double *rhos, lfdmi_result *results, int loc) {
    if (!rhos) return fail(ctx, LFDMI_ERR_ARG, "rhos NULL");
    int mode = dim ? (after_bright ? LFDMI_PREP_BRIGHT_THEN_DIM : LFDMI_PREP_DIM) : LFDMI_PREP_BRIGHT;
    return pass_api(ctx, img, dtype, n, h, w, flip, mode, dim != 0, p, n_scales, rhos, results, (size_t)n, nullptr, nullptr, loc);
}

// ---- host-frame feed -------------------------------------------------------------------------------------------------
// A pageable hipMemcpyAsync reaches ~30-45 GB/s on this platform and cannot overlap with anything (the runtime stages it
// synchronously); pinning the caller's array costs 40 ms per GB.  So the library stages itself: `feed_threads` host
// threads copy chunk k+1 of the caller's frames into a pinned buffer (~100 GB/s with 8 threads) while the DMA engine
// moves chunk k (57 GB/s, the link's rate) and the GPU processes chunk k-1: the frames cross PCIe once, back to back.
// One feeder thread per call walks the chunks IN ORDER: host threads copy a piece (32 MB) of the caller's frames into the
// pinned buffer of the chunk's slot, the piece is sent on its way at once (copy stream), and after the chunk's last piece an
// event is recorded.  The DMA engine therefore starts after the first piece (< 1 ms) and never sees two chunks interleaved.
// Chunk j reuses the buffers of chunk j - 2: the feeder waits until the main thread has finished that chunk (`done`).
// All waiting is on one condition variable (no spinning: eight ranks on one host would burn 40 cores), the workers look at
// `stop` before every piece, and the first HIP error of an upload ends the feed and is handed to the caller by feed_wait.
#define FEED_PIECE (32u << 20)
struct FeedState {
    std::mutex mu;
    std::condition_variable cv;
    int issued = 0, done = 0;          // chunks whose upload has been enqueued / that the main thread has finished (guarded by mu)
    bool stop = false;
    hipError_t err = hipSuccess;       // first failure of an upload
    std::thread th;
};

// CPUs next to the GPU (its PCI function's local_cpulist) that this process may run on; empty when unknown or LFDMI_NUMA_PIN=0.
// Feed / blot threads and the pinned staging buffers are bound to them: with one rank per GPU on a two-socket host a rank's
// 50 GB/s of staging copies otherwise cross the socket link at the scheduler's whim.
static std::vector<int> numa_cpus(int device) {
    std::vector<int> out;
    if (const char *e = getenv("LFDMI_NUMA_PIN")) if (!atoi(e)) return out;
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, device) != hipSuccess) return out;
    for (char *c = bus; *c; c++) if (*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a');
    std::string path = std::string("/sys/bus/pci/devices/") + bus + "/local_cpulist";
    FILE *f = fopen(path.c_str(), "r");
    if (!f) return out;
    char line[4096] = {0};
    const bool got = fgets(line, sizeof line, f) != nullptr;
    fclose(f);
    if (!got) return out;
    cpu_set_t allowed;
    CPU_ZERO(&allowed);
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return out;
    for (char *tok = strtok(line, ",\n"); tok; tok = strtok(nullptr, ",\n")) { // "0-15,128-143"
        int a = 0, b = 0;
        const int k = sscanf(tok, "%d-%d", &a, &b);
        if (k < 1) continue;
        if (k == 1) b = a;
        for (int c = a; c <= b && c < CPU_SETSIZE; c++) if (c >= 0 && CPU_ISSET(c, &allowed)) out.push_back(c);
    }
    return out;
}
static void bind_thread(const std::vector<int> &cpus) { // the calling thread; no-op for an empty list
    if (cpus.empty()) return;
    cpu_set_t set;
    CPU_ZERO(&set);
    for (int c : cpus) CPU_SET(c, &set);
    (void)pthread_setaffinity_np(pthread_self(), sizeof set, &set);
}

static int feed_prepare(lfdmi_ctx *ctx, size_t bytes, bool need_pin) {
    if (!ctx->feed_copy) {
        HIPCHK(hipStreamCreateWithFlags(&ctx->feed_copy, hipStreamNonBlocking));
        HIPCHK(hipStreamCreateWithFlags(&ctx->feed_copy2, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&ctx->feed_mid, hipEventDisableTiming));
        for (int i = 0; i < 2; i++) HIPCHK(hipEventCreateWithFlags(&ctx->feed_up[i], hipEventDisableTiming));
    }
    if (!ctx->feed_cpus_known) { ctx->feed_cpus = numa_cpus(ctx->device); ctx->feed_cpus_known = true; }
    if (ctx->feed_bytes >= bytes && (!need_pin || ctx->feed_pin_bytes >= bytes)) return 0;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->feed_copy));
    HIPCHK(hipStreamSynchronize(ctx->feed_copy2));
    if (ctx->feed_bytes < bytes) {
        for (int i = 0; i < 2; i++)
            if (ctx->feed_dev[i]) { HIPCHK(hipFree(ctx->feed_dev[i])); ctx->feed_dev[i] = nullptr; }
        ctx->feed_bytes = 0;
        for (int i = 0; i < 2; i++) HIPCHK(hipMalloc(&ctx->feed_dev[i], bytes));
        ctx->feed_bytes = bytes;
    }
    if (need_pin && ctx->feed_pin_bytes < bytes) {
        for (int i = 0; i < 2; i++)
            if (ctx->feed_pin[i]) { HIPCHK(hipHostFree(ctx->feed_pin[i])); ctx->feed_pin[i] = nullptr; }
        ctx->feed_pin_bytes = 0;
        // the pinned buffers are allocated (and first touched) by a thread bound to the GPU's CPUs: their pages land on that node
        hipError_t herr = hipSuccess;
        std::thread alloc([&] {
            bind_thread(ctx->feed_cpus);
            herr = hipSetDevice(ctx->device);
            for (int i = 0; i < 2 && herr == hipSuccess; i++) {
                herr = hipHostMalloc(&ctx->feed_pin[i], bytes, hipHostMallocDefault);
                if (herr == hipSuccess && !ctx->feed_cpus.empty())
                    for (size_t o = 0; o < bytes; o += 4096) ((volatile char *)ctx->feed_pin[i])[o] = 0;
            }
        });
        alloc.join();
        HIPCHK(herr);
        ctx->feed_pin_bytes = bytes;
    }
    return 0;
}

static void feed_start(lfdmi_ctx *ctx, FeedState *fs, const char *src, size_t frame_bytes, int n, int per) {
    char *pin[2] = {(char *)ctx->feed_pin[0], (char *)ctx->feed_pin[1]}, *dev[2] = {(char *)ctx->feed_dev[0], (char *)ctx->feed_dev[1]};
    hipEvent_t up[2] = {ctx->feed_up[0], ctx->feed_up[1]};
    const int T = ctx->feed_threads, device = ctx->device;
    hipStream_t copy = ctx->feed_copy, copy2 = ctx->feed_streams == 2 ? ctx->feed_copy2 : ctx->feed_copy;
    hipEvent_t mid = ctx->feed_mid;
    const std::vector<int> cpus = ctx->feed_cpus;
    fs->th = std::thread([=] {
        // The call's pieces in order; T workers (this thread is one of them) live for the whole call and copy their slice
        // of every piece (threads started per piece cost as much as the copy itself); whoever completes a piece sends it
        // -- and any completed pieces queued behind an unfinished one -- on its way, strictly in order.
        struct Piece { int chunk; size_t src_off, off, bytes; bool last; };
        std::vector<Piece> pieces;
        {
            int j = 0;
            for (int c0 = 0; c0 < n; c0 += per, j++) {
                const size_t bytes = (size_t)std::min(per, n - c0) * frame_bytes;
                for (size_t o = 0; o < bytes; o += FEED_PIECE)
                    pieces.push_back({j, (size_t)c0 * frame_bytes, o, std::min<size_t>(FEED_PIECE, bytes - o), o + FEED_PIECE >= bytes});
            }
        }
        const int P = (int)pieces.size();
        std::unique_ptr<std::atomic<int>[]> copied(new std::atomic<int>[P]);
        for (int k = 0; k < P; k++) copied[k].store(0);
        int next_issue = 0; // (guarded by fs->mu)
        auto worker = [&](int t) {
            bind_thread(cpus);
            if (hipSetDevice(device) != hipSuccess) {
                std::lock_guard<std::mutex> lk(fs->mu);
                if (fs->err == hipSuccess) fs->err = hipErrorInvalidDevice;
                fs->stop = true;
                fs->cv.notify_all();
                return;
            }
            for (int k = 0; k < P; k++) {
                const Piece &pc = pieces[k];
                {   // buffers of chunk j - 2 still in use?  `stop` is looked at before every piece, not only while waiting
                    std::unique_lock<std::mutex> lk(fs->mu);
                    fs->cv.wait(lk, [&] { return fs->stop || fs->done >= pc.chunk - 1; });
                    if (fs->stop) return;
                }
                const int slot = pc.chunk & 1;
                const size_t a = (pc.bytes * t / T) & ~(size_t)63, b = t == T - 1 ? pc.bytes : ((pc.bytes * (t + 1) / T) & ~(size_t)63);
                if (b > a) memcpy(pin[slot] + pc.off + a, src + pc.src_off + pc.off + a, b - a);
                if (copied[k].fetch_add(1, std::memory_order_acq_rel) + 1 == T) {
                    std::lock_guard<std::mutex> lk(fs->mu);
                    bool published = false;
                    while (!fs->stop && next_issue < P && copied[next_issue].load(std::memory_order_acquire) == T) {
                        const Piece &q = pieces[next_issue];
                        const int qs = q.chunk & 1;
                        hipError_t e = hipMemcpyAsync(dev[qs] + q.off, pin[qs] + q.off, q.bytes, hipMemcpyHostToDevice, (next_issue & 1) ? copy2 : copy);
                        if (e == hipSuccess && q.last) {
                            if (copy2 != copy) {
                                e = hipEventRecord(mid, copy2);
                                if (e == hipSuccess) e = hipStreamWaitEvent(copy, mid, 0);
                            }
                            if (e == hipSuccess) e = hipEventRecord(up[qs], copy);
                            if (e == hipSuccess) { fs->issued = q.chunk + 1; published = true; }
                        }
                        if (e != hipSuccess) { // the caller gets LFDMI_ERR_HIP from feed_wait; nothing further is uploaded
                            fs->err = e;
                            fs->stop = true;
                            published = true;
                            break;
                        }
                        next_issue++;
                    }
                    if (published) fs->cv.notify_all();
                }
            }
        };
        std::vector<std::thread> th;
        for (int t = 1; t < T; t++) th.emplace_back(worker, t);
        worker(0);
        for (auto &x : th) x.join();
    });
}

static double feed_now() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}
// the launch stream waits for chunk j's upload
static int feed_wait(lfdmi_ctx *ctx, FeedState *fs, int j) {
    {
        std::unique_lock<std::mutex> lk(fs->mu);
        fs->cv.wait(lk, [&] { return fs->issued > j || fs->stop; });
        if (fs->issued <= j) {
            const hipError_t e = fs->err;
            return fail(ctx, LFDMI_ERR_HIP, std::string("host-frame feed: ") + (e != hipSuccess ? hipGetErrorString(e) : "stopped"));
        }
    }
    HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->feed_up[j & 1], 0));
    return 0;
}
static void feed_done(FeedState *fs, int chunks_done) { // chunk (chunks_done - 1)'s buffers are free again
    std::lock_guard<std::mutex> lk(fs->mu);
    fs->done = chunks_done;
    fs->cv.notify_all();
}

// big-endian float32 -> native, in place (the raw data unit of a BITPIX = -32 FITS image, detecttrails.py:113): 16 bytes per lane
__global__ void __launch_bounds__(256) k_bswap32(uint4 *p, size_t n16) {
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n16; k += (size_t)gridDim.x * 256) {
        uint4 v = p[k];
        v.x = __builtin_bswap32(v.x); v.y = __builtin_bswap32(v.y); v.z = __builtin_bswap32(v.z); v.w = __builtin_bswap32(v.w);
        p[k] = v;
    }
}
static int run_bswap(lfdmi_ctx *ctx, void *dev, size_t bytes) { // bytes % 16 == 0 is not required: the tail is swapped by one more wave
    Span sp(ctx, KID_MISC);
    const size_t n16 = bytes / 16;
    if (n16) {
        k_bswap32<<<(unsigned)std::min<size_t>((n16 + 255) / 256, 16384), 256, 0, ctx->stream>>>((uint4 *)dev, n16);
        KCHK("k_bswap32");
    }
    if (bytes % 16) { // (frames of h * w floats: only when h * w is not a multiple of 4)
        std::vector<uint32_t> tail((bytes % 16) / 4);
        char *q = (char *)dev + n16 * 16;
        HIPCHK(hipMemcpyAsync(tail.data(), q, tail.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        for (auto &v : tail) v = __builtin_bswap32(v);
        HIPCHK(hipMemcpyAsync(q, tail.data(), tail.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    return 0;
}

// pinned frames (LFDMI_HOST_PINNED): chunk kc's bytes go from the caller's buffer to the device in 32 MB pieces alternating
// between the two copy streams, no staging copy; the event of the chunk's slot is recorded behind the last piece
static int pinned_upload(lfdmi_ctx *ctx, const char *src, size_t bytes, int kc) {
    char *dst = (char *)ctx->feed_dev[kc & 1];
    hipStream_t copy = ctx->feed_copy, copy2 = ctx->feed_streams == 2 ? ctx->feed_copy2 : ctx->feed_copy;
    int k = 0;
    for (size_t o = 0; o < bytes; o += FEED_PIECE, k++)
        HIPCHK(hipMemcpyAsync(dst + o, src + o, std::min<size_t>(FEED_PIECE, bytes - o), hipMemcpyHostToDevice, (k & 1) ? copy2 : copy));
    if (copy2 != copy) {
        HIPCHK(hipEventRecord(ctx->feed_mid, copy2));
        HIPCHK(hipStreamWaitEvent(copy, ctx->feed_mid, 0));
    }
    HIPCHK(hipEventRecord(ctx->feed_up[kc & 1], copy));
    return 0;
}

static int detect_impl(lfdmi_ctx *ctx, void *frames_v, int dtype, int n, int h, int w, const lfdmi_catalog *cat,
                       const lfdmi_rs_params *rs, const lfdmi_params *bright, const lfdmi_params *dim,
                       lfdmi_result *results, int loc, int cat_f0 = 0) { // cat_f0: catalogue entry of frame 0 (a frame run again alone)
    RET(check_shape(ctx, n, h, w));
    RET(check_params(ctx, bright, false));
    RET(check_params(ctx, dim, true));
    float *frames = (float *)frames_v;
    if (!frames || !results) return fail(ctx, LFDMI_ERR_ARG, "NULL argument");
    if (dtype != LFDMI_F32 && dtype != LFDMI_F32_BE) return fail(ctx, LFDMI_ERR_DTYPE, "lfdmi_detect_batch_raw: frames must be LFDMI_F32 or LFDMI_F32_BE");
    if (loc != LFDMI_HOST && loc != LFDMI_DEVICE && loc != LFDMI_HOST_PINNED) return fail(ctx, LFDMI_ERR_ARG, "bad loc");
    if (dtype == LFDMI_F32_BE && loc == LFDMI_DEVICE) {
        // big-endian frames already on the device (decoded there: lfdmi_bz2_*): swapped in place, once -- a chunk that is run again
        // (table growth, general path) must not be swapped again -- and native from here on
        HIPCHK(hipSetDevice(ctx->device));
        RET(run_bswap(ctx, frames, (size_t)n * h * w * 4));
        dtype = LFDMI_F32;
    }
    const bool be = dtype == LFDMI_F32_BE;
    const bool pinned = loc == LFDMI_HOST_PINNED;
    if (pinned) loc = LFDMI_HOST; // (everything below but the upload treats them as host frames)
    if (!hough_fits(ctx, h, w, bright->houghMethod, LFD_PI / 180) || !hough_fits(ctx, h, w, dim->houghMethod, LFD_PI / 180)) {
        lfdmi_ctx *sp = get_spill(ctx); // rho finer than this workspace's accumulators: the worst-case one takes the call
        if (!sp || !hough_fits(sp, h, w, bright->houghMethod, LFD_PI / 180) || !hough_fits(sp, h, w, dim->houghMethod, LFD_PI / 180))
            return fail(ctx, LFDMI_ERR_CAPACITY, "Hough accumulator larger than the workspace (rho < 1 px)");
        int rc = detect_impl(sp, frames, dtype, n, h, w, cat, rs, bright, dim, results, pinned ? LFDMI_HOST_PINNED : loc, cat_f0);
        if (rc) ctx->err = sp->err;
        return rc;
    }
    size_t N = (size_t)h * w;
    int *const flags = (int *)ctx->res_host;
    lfdmi_result *const host = (lfdmi_result *)((char *)ctx->res_host + (size_t)ctx->G * sizeof(int));
    std::vector<int4> boxes;
    KeepEqu keep_guard(ctx, ctx->stage_mode == 1);
    ctx->stages_valid = ctx->keep_equ;
    // both passes read the same float frames: one sweep feeds both where the dim pass's front end can be fused at all
    const bool dual = ctx->fuse_dual && can_fuse_prep_erode(ctx, LFDMI_F32, w, dim->erodeKernel, dim->erode_kh, dim->erode_kw);
    // ... or, cheaper, the bright pass's sweep leaves one bit per pixel from which the dim pass rebuilds its 8-bit image
    // (dim value = bright value + bit, for 0 <= addFlux <= 1 and minFlux <= 0.5) together with that image's histogram
    // (small erosion kernels only: k_bits_erode fetches kh x kw values per surviving pixel, which is nothing on sky frames but
    // would be slow on a dense image with a large kernel; those keep the band kernel)
    // (addFlux must stay clear of 1: x = 0.5 gives bright rne(0.5) = 0 but dim rne(1.5) = 2, and an addFlux within half an ulp
    // of 1 rounds x + addFlux up to the same tie; up to 0.999 the float sum of an even k + 0.5 < 256 stays below k + 1.5)
    const bool delta = !dual && ctx->delta_dim && (w % 32) == 0 && (float)dim->addFlux >= 0.0f && (float)dim->addFlux <= 0.999f &&
                       (float)dim->minFlux <= 0.5f &&
                       dim->erode_kh * dim->erode_kw <= 25 &&
                       can_fuse_prep_erode(ctx, LFDMI_F32, w, dim->erodeKernel, dim->erode_kh, dim->erode_kw);
    // Host frames: chunks of up to ~feed_chunk_bytes (and at most G frames) go through the pinned double buffer
    // (feed_* above), chunk k+1 uploading while chunk k is processed.  Device frames (LFDMI_FEED_MB=0, tiny batches):
    // chunks of G frames, used in place / staged by the runtime.
    int per = ctx->G;
    const bool feed = !pinned && loc == LFDMI_HOST && ctx->feed_chunk_bytes > 0 && (size_t)n * N * 4 >= (64u << 20);
    if (feed || pinned) {
        size_t fpc = (ctx->feed_chunk_bytes ? ctx->feed_chunk_bytes : (size_t)(800u << 20)) / (N * 4);
        per = (int)std::min<size_t>((size_t)ctx->G, std::max<size_t>(1, fpc));
        // pinned frames: one upload per call unless asked otherwise.  Cutting a call into k upload chunks (LFDMI_PINNED_CHUNKS=k,
        // developer knob, read per call) overlaps chunk c + 1's upload with chunk c's passes, but a 64-frame call is 14 ms of PCIe
        // for 1.5 ms of kernels, and what the overlap buys depends on which hardware queues HIP happens to give the five streams
        // of a context: measured 2 940 - 3 650 frames/s with k = 4 across process states (torch initialised or not, a second
        // context alive, GPU_MAX_HW_QUEUES=8, high-priority copy streams) against 3 430 - 3 500 in all of them with k = 1
        // (profiles/README.md, round-3 log)
        if (pinned) {
            const char *pe = getenv("LFDMI_PINNED_CHUNKS");
            const int pin_div = pe ? std::max(1, atoi(pe)) : 1;
            per = std::max(std::min(per, 8), std::min(per, (n + pin_div - 1) / pin_div));
        }
        RET(feed_prepare(ctx, (size_t)std::min(per, n) * N * 4, feed));
    }
    FeedState fs;
    std::thread blotter; // remove_stars on the caller's host array (see blot_host_frames)
    struct FeedJoin { // every exit path: the feeder stops before its next piece, uploads still in flight are drained (they read the
                      // pinned buffers the next call refills), the blotter is joined
        lfdmi_ctx *c; FeedState *f; std::thread *b; bool feed;
        ~FeedJoin() {
            { std::lock_guard<std::mutex> lk(f->mu); f->stop = true; f->cv.notify_all(); }
            if (f->th.joinable()) f->th.join();
            if (feed) { (void)hipStreamSynchronize(c->feed_copy); (void)hipStreamSynchronize(c->feed_copy2); }
            if (b->joinable()) b->join();
        }
    } feed_join{ctx, &fs, &blotter, feed || pinned};
    if (feed) feed_start(ctx, &fs, (const char *)frames, N * 4, n, per);
    if (pinned && n > 0) RET(pinned_upload(ctx, (const char *)frames, (size_t)std::min(per, n) * N * 4, 0));
    const int fail_chunk = ctx->fail_chunk;
    ctx->fail_chunk = -1;
    for (int c0 = 0, kc = 0; c0 < n; c0 += per, kc++) {
        int nc = n - c0 < per ? n - c0 : per;
        const void *d;
        if (kc == fail_chunk) return fail(ctx, LFDMI_ERR_ARG, "lfdmi_debug_fail_chunk: injected failure");
        if (feed) {
            d = ctx->feed_dev[kc & 1];
            RET(feed_wait(ctx, &fs, kc)); // the launch stream waits for chunk kc's upload (chunk kc + 1 follows it back to back)
        } else if (pinned) {
            // chunk kc + 1 goes on its way before chunk kc's kernels are enqueued (its device buffer was chunk kc - 1's, whose
            // passes have been synchronised), so the DMA engines and the compute units overlap
            d = ctx->feed_dev[kc & 1];
            if (c0 + per < n)
                RET(pinned_upload(ctx, (const char *)frames + (size_t)(c0 + per) * N * 4, (size_t)std::min(per, n - c0 - per) * N * 4, kc + 1));
            HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->feed_up[kc & 1], 0));
        } else RET(in_ptr(ctx, frames, (size_t)c0 * N * 4, (size_t)nc * N * 4, loc, &d));
        if (be) RET(run_bswap(ctx, (void *)d, (size_t)nc * N * 4));
        // (raw big-endian frames are a file's data unit, a read-only input: only the device copy is blotted -- no squares come
        // back, no mid-call synchronisation, no host threads)
        const bool host_blot = cat && loc == LFDMI_HOST && cat->loc == LFDMI_HOST && !be;
        // The bit-plane sweep can mask remove_stars' squares as it loads the values (k_prep_hist<.., RS>), so nothing waits for the
        // zero fill: a host frame's device copy is never filled at all (the caller's array is blotted by host threads, a big-endian
        // frame is read-only), a device-resident frame -- which the caller does get back blotted -- is filled on a side stream
        // beside the latency-bound stages of the bright pass (rs_fill_point).  Frames whose blotted copy travels back over PCIe
        // (catalogue on the device, frames on the host) keep the fill in front.
        const bool copy_back = cat && loc == LFDMI_HOST && !host_blot && !be;
        ctx->rs_fold = cat && cat->max_obj > 0 && ctx->rs_fold_on && delta && w <= RS_MAXW && !copy_back;
        ctx->rs_fill_frames = nullptr;
        struct FoldState { // (an error return in the middle of a chunk must not leave the fill running on the caller's frames)
            lfdmi_ctx *c;
            ~FoldState() {
                if (c->rs_fill_inflight) { (void)hipStreamSynchronize(c->side[0]); (void)hipStreamSynchronize(c->side[1]); c->rs_fill_inflight = false; }
                c->rs_fold = false; c->rs_fill_frames = nullptr; c->rs_fill_part = 0;
            }
        } fold_guard{ctx};
        if (cat) {
            RET(run_removestars(ctx, (float *)d, cat_f0 + c0, nc, h, w, cat, rs, host_blot ? &boxes : nullptr, !ctx->rs_fold));
            if (ctx->rs_fold && loc == LFDMI_DEVICE) { ctx->rs_fill_frames = (float *)d; ctx->rs_fill_part = 0; ctx->rs_fill_nc = nc; ctx->rs_fill_h = h; ctx->rs_fill_w = w; }
            if (copy_back) RET(out_copy(ctx, frames, (size_t)c0 * N * 4, d, (size_t)nc * N * 4, loc));
            if (host_blot) {
                double t0_ = feed && getenv("LFDMI_FEED_TRACE") ? feed_now() : 0;
                HIPCHK(hipStreamSynchronize(ctx->stream)); // the squares are on the host; the passes are enqueued next
                if (t0_ > 0) fprintf(stderr, "[feed] chunk %d: upload + remove_stars synced after %.2f ms (t=%.2f)\n", kc, feed_now() - t0_, feed_now());
            }
        }
        bool blotted = false;
        GeneralGuard gg(ctx);
        int grow_tries = 0;
        for (;;) { // (again, with the general run kernels, if a frame turned out to need them)
        k_init_results<<<(nc + 63) / 64, 64, 0, ctx->stream>>>(ctx->res_dev, ctx->pass_flags, nc);
        KCHK("k_init_results");
        struct DualState { lfdmi_ctx *c; ~DualState() { c->dual_state = 0; } } dual_guard{ctx};
        struct DeltaState { lfdmi_ctx *c; ~DeltaState() { c->delta_state = 0; } } delta_guard{ctx};
        ctx->cur_pass = 0;
        ctx->dual_state = 0;
        ctx->delta_state = delta ? 1 : 0;
        RET(run_pass(ctx, d, LFDMI_F32, nc, h, w, 1, LFDMI_PREP_BRIGHT, false, bright, nullptr, ctx->need_dim, (dual || delta) ? dim : nullptr));
        ctx->cur_pass = 1;
        ctx->dual_state = dual ? 2 : 0;
        ctx->delta_state = delta ? 2 : 0;
        RET(run_pass(ctx, d, LFDMI_F32, nc, h, w, 1, LFDMI_PREP_BRIGHT_THEN_DIM, true, dim, ctx->need_dim, nullptr));
        ctx->cur_pass = 0;
        ctx->dual_state = 0;
        ctx->delta_state = 0;
        if (host_blot && !blotted) { // host threads zero-fill the caller's frames in the background (joined below / at the end)
            if (blotter.joinable()) blotter.join();
            blotter = std::thread([=, bx = boxes, cpus = ctx->feed_cpus] { blot_host_frames(frames + (size_t)c0 * N, nc, h, w, cat, cat_f0 + c0, bx, cpus); });
            blotted = true;
        }
        RET(rs_fill_point(ctx, -1)); // (a fill no stage has started yet)
        if (ctx->rs_fill_inflight) { HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_rsfill, 0)); ctx->rs_fill_inflight = false; }
        HIPCHK(hipMemcpyAsync(ctx->res_host, ctx->pass_flags, (size_t)ctx->G * sizeof(int) + (size_t)nc * sizeof(lfdmi_result), hipMemcpyDeviceToHost, ctx->stream));
        { double t0_ = feed && getenv("LFDMI_FEED_TRACE") ? feed_now() : 0;
        HIPCHK(hipStreamSynchronize(ctx->stream));
        if (t0_ > 0) fprintf(stderr, "[feed] chunk %d (%d frames): passes synced after %.2f ms (t=%.2f)\n", kc, nc, feed_now() - t0_, feed_now()); }
        { // the look-back scan gave up on a frame (the GPU is shared with another process: see k_scan_fused): three launches from
          // now on, and this chunk once more (cheaper than the worst-case rerun of every frame that was flagged)
            if (scan_gave_up(ctx, flags, nc)) continue;
        }
        if (grow_tries < 4 && grow_caps(ctx, host, nc)) { grow_tries++; continue; } // tables enlarged for this chunk's frames: once more
        if (!gg.again(flags, nc)) break;
        }
        chunk_done(ctx);
        {
            int na[2] = {nc, 0}, nd[2] = {0, 0};
            for (int i = 0; i < nc; i++) { nd[0] += flags[i] & 1; na[1] += (flags[i] >> 1) & 1; nd[1] += (flags[i] >> 2) & 1; }
            RET(sync(ctx, nc, na, nd));
        }
        if (feed) feed_done(&fs, kc + 1); // this chunk's two buffers are free again: chunk kc + 2 may start
        for (int i = 0; i < nc; i++) {
            if (host[i].status == LFDMI_ERR_CAPACITY) {
                // a table of this workspace was too small for the frame: once more, alone, in the worst-case workspace.
                // remove_stars has already blotted the frame (device copy and, by now, the caller's array): no catalogue.
                lfdmi_ctx *sp = get_spill(ctx);
                if (sp) {
                    if (blotter.joinable()) blotter.join(); // (a host frame is read again below: its blotting must be complete)
                    // (a big-endian frame is a read-only input: what the rerun uploads is not blotted, it needs the frame's catalogue entry)
                    int rc = detect_impl(sp, frames + (size_t)(c0 + i) * N, dtype, 1, h, w, be ? cat : nullptr, be ? rs : nullptr, bright, dim,
                                         &results[c0 + i], loc, cat_f0 + c0 + i);
                    if (rc) { ctx->err = sp->err; return rc; }
                    ctx->n_spilled++;
                    continue;
                }
            }
            dictify(h, w, &host[i]);
            results[c0 + i] = host[i];
        }
    }
    return 0;
}

extern "C" int lfdmi_detect_batch(lfdmi_ctx *ctx, float *frames, int n, int h, int w, const lfdmi_catalog *cat,
                                  const lfdmi_rs_params *rs, const lfdmi_params *bright, const lfdmi_params *dim,
                                  lfdmi_result *results, int loc) {
    return detect_impl(ctx, frames, LFDMI_F32, n, h, w, cat, rs, bright, dim, results, loc);
}
extern "C" int lfdmi_detect_batch_raw(lfdmi_ctx *ctx, void *frames, int dtype, int n, int h, int w, const lfdmi_catalog *cat,
                                      const lfdmi_rs_params *rs, const lfdmi_params *bright, const lfdmi_params *dim,
                                      lfdmi_result *results, int loc) {
    return detect_impl(ctx, frames, dtype, n, h, w, cat, rs, bright, dim, results, loc);
}

// pinned host memory for LFDMI_HOST_PINNED frames: allocated (and first touched) by a thread bound to the CPUs next to the GPU
extern "C" int lfdmi_host_alloc(lfdmi_ctx *ctx, uint64_t bytes, void **out) {
    if (!ctx || !out || bytes == 0) return fail(ctx, LFDMI_ERR_ARG, "lfdmi_host_alloc: bad argument");
    *out = nullptr;
    if (!ctx->feed_cpus_known) { ctx->feed_cpus = numa_cpus(ctx->device); ctx->feed_cpus_known = true; }
    hipError_t herr = hipSuccess;
    void *p = nullptr;
    std::thread alloc([&] {
        bind_thread(ctx->feed_cpus);
        herr = hipSetDevice(ctx->device);
        if (herr == hipSuccess) herr = hipHostMalloc(&p, (size_t)bytes, hipHostMallocDefault);
        if (herr == hipSuccess && !ctx->feed_cpus.empty())
            for (size_t o = 0; o < (size_t)bytes; o += 4096) ((volatile char *)p)[o] = 0;
    });
    alloc.join();
    HIPCHK(herr);
    *out = p;
    return 0;
}
extern "C" int lfdmi_host_free(lfdmi_ctx *ctx, void *p) {
    if (!p) return 0;
    if (ctx) (void)hipSetDevice(ctx->device);
    hipError_t e = hipHostFree(p);
    if (e != hipSuccess) return fail(ctx, LFDMI_ERR_HIP, std::string("hipHostFree: ") + hipGetErrorString(e));
    return 0;
}

extern "C" int lfdmi_get_counters(lfdmi_ctx *ctx, int slot0, int n, int32_t *dst) {
    if (!ctx || !dst || slot0 < 0 || n < 0 || slot0 + n > ctx->G) return LFDMI_ERR_ARG;
    static_assert(C_COUNT == LFDMI_COUNTERS, "counter layout");
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipMemcpyAsync(dst, ctx->counters + (size_t)slot0 * C_COUNT, (size_t)n * C_COUNT * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
}

// developer hook (tests of the error path): the next lfdmi_detect_batch call on this context returns LFDMI_ERR_ARG at the top
// of chunk `chunk` (chunks are feed-sized for host frames, max_inflight frames otherwise); -1 disarms
extern "C" int lfdmi_debug_fail_chunk(lfdmi_ctx *ctx, int chunk) {
    if (!ctx) return LFDMI_ERR_ARG;
    ctx->fail_chunk = chunk;
    return 0;
}

// developer tool: phase clocks of the last k_frame_contours launch
extern "C" int lfdmi_debug_frame_profile(lfdmi_ctx *ctx, int n, long long *dst) {
    if (!ctx || !ctx->prof || n < 0 || n > ctx->G) return LFDMI_ERR_ARG;
    HIPCHK(hipMemcpyAsync(dst, ctx->prof, (size_t)n * 16 * sizeof(long long), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
}

// developer / test entry point: the tail of a pass on line sets handed in from outside -- k_finalize (the device's check_theta,
// processfield.py:36-150, with the zero fill of processfield.py:89-102) and the host-side dictify_hough (processfield.py:266-288)
// exactly as lfdmi_detect_batch runs them.  h1 / h2: [n][kmax][2] float32 (rho, theta), n1 / n2: lines per set (<= kmax);
// out[i].rejected_by_theta = check_theta's True, .found = `which` when it returns None, then rho / theta / x1 .. y2.
extern "C" int lfdmi_debug_tail(lfdmi_ctx *ctx, int n, int kmax, const float *h1, const int32_t *n1, const float *h2, const int32_t *n2,
                                int navg, double dro, double thetaTresh, double lineSetTresh, int which, int h, int w, lfdmi_result *out) {
    if (!ctx || n < 0 || !h1 || !h2 || !n1 || !n2 || !out || kmax < 1 || navg < 1 || navg > LFDMI_MAX_SET_LINES || (which != 1 && which != 2))
        return fail(ctx, LFDMI_ERR_ARG, "lfdmi_debug_tail: bad argument");
    HIPCHK(hipSetDevice(ctx->device));
    const int G = ctx->G, K = navg;
    std::vector<float> lines((size_t)G * 2 * K * 2);
    std::vector<int> cnt((size_t)G * C_COUNT);
    for (int c0 = 0; c0 < n; c0 += G) {
        const int nc = std::min(G, n - c0);
        std::fill(lines.begin(), lines.end(), 0.f);
        std::fill(cnt.begin(), cnt.end(), 0);
        for (int i = 0; i < nc; i++) {
            const int a = n1[c0 + i], b = n2[c0 + i];
            if (a < 0 || a > kmax || b < 0 || b > kmax) return fail(ctx, LFDMI_ERR_ARG, "lfdmi_debug_tail: line count");
            // (what k_hough_topk leaves: the first K lines of each set, zero-filled)
            for (int k = 0; k < K && k < a; k++) { lines[((size_t)i * 2 + 0) * K * 2 + 2 * k] = h1[((size_t)(c0 + i) * kmax + k) * 2]; lines[((size_t)i * 2 + 0) * K * 2 + 2 * k + 1] = h1[((size_t)(c0 + i) * kmax + k) * 2 + 1]; }
            for (int k = 0; k < K && k < b; k++) { lines[((size_t)i * 2 + 1) * K * 2 + 2 * k] = h2[((size_t)(c0 + i) * kmax + k) * 2]; lines[((size_t)i * 2 + 1) * K * 2 + 2 * k + 1] = h2[((size_t)(c0 + i) * kmax + k) * 2 + 1]; }
            cnt[(size_t)i * C_COUNT + C_DETECT] = 1;
            cnt[(size_t)i * C_COUNT + C_NPEAK_EQU] = a;
            cnt[(size_t)i * C_COUNT + C_NPEAK_BOX] = b;
        }
        HIPCHK(hipMemcpyAsync(ctx->lines, lines.data(), (size_t)nc * 2 * K * 2 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(ctx->counters, cnt.data(), (size_t)nc * C_COUNT * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
        k_init_results<<<(nc + 63) / 64, 64, 0, ctx->stream>>>(ctx->res_dev, ctx->pass_flags, nc);
        KCHK("k_init_results");
        TailParams tp;
        tp.navg = navg; tp.dro = dro; tp.thetaTresh = thetaTresh; tp.lineSetTresh = lineSetTresh; tp.which = which;
        k_finalize<<<(nc + 63) / 64, 64, 0, ctx->stream>>>(ctx->lines, ctx->counters, ctx->res_dev, nullptr, ctx->pass_flags, nullptr, tp, nc);
        KCHK("k_finalize");
        HIPCHK(hipMemcpyAsync(out + c0, ctx->res_dev, (size_t)nc * sizeof(lfdmi_result), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        for (int i = 0; i < nc; i++) dictify(h, w, &out[c0 + i]);
    }
    return 0;
}

extern "C" int lfdmi_debug_trig(lfdmi_ctx *ctx, int n, const double *y, const double *x, float *angle_deg, float *cos_half, float *sin_half) {
    if (!ctx || n < 0 || !y || !x || !angle_deg || !cos_half || !sin_half) return LFDMI_ERR_ARG;
    if (n == 0) return 0;
    HIPCHK(hipSetDevice(ctx->device));
    RET(ensure_scratch(ctx, (size_t)n * (2 * sizeof(double) + 3 * sizeof(float))));
    double *dy = (double *)ctx->scratch, *dx = dy + n;
    float *o = (float *)(dx + n);
    HIPCHK(hipMemcpyAsync(dy, y, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(dx, x, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    k_debug_trig<<<(n + 255) / 256, 256, 0, ctx->stream>>>(dy, dx, o, o + n, o + 2 * (size_t)n, n);
    KCHK("k_debug_trig");
    HIPCHK(hipMemcpyAsync(angle_deg, o, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(cos_half, o + n, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(sin_half, o + 2 * (size_t)n, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
}

extern "C" int lfdmi_set_stage_images(lfdmi_ctx *ctx, int mode) {
    if (!ctx || mode < -1 || mode > 1) return LFDMI_ERR_ARG;
    ctx->stage_mode = mode;
    return 0;
}

extern "C" int lfdmi_get_stage(lfdmi_ctx *ctx, int slot, int which, int h, int w, uint8_t *dst, int loc) {
    if (!ctx || !dst || slot < 0 || slot >= ctx->G) return LFDMI_ERR_ARG;
    if (ctx->last_h <= 0) return fail(ctx, LFDMI_ERR_ARG, "no call has run yet");
    if (h != ctx->last_h || w != ctx->last_w) return fail(ctx, LFDMI_ERR_ARG, "lfdmi_get_stage: the last call worked on a different shape");
    size_t N = (size_t)h * w, BW = (size_t)h * LFD_WQ(w);
    HIPCHK(hipSetDevice(ctx->device));
    const uint8_t *src = nullptr;
    if (which != LFDMI_STAGE_CANNY && which != LFDMI_STAGE_BOX && !ctx->stages_valid && !(which == LFDMI_STAGE_ERODED && ctx->eroded_valid))
        return fail(ctx, LFDMI_ERR_ARG, "lfdmi_get_stage: the last call did not keep the 8-bit stage images (lfdmi_set_stage_images)");
    if (which == LFDMI_STAGE_GRAY) src = ctx->gray + slot * N;
    else if (which == LFDMI_STAGE_EQU) src = ctx->equ + slot * N;
    else if (which == LFDMI_STAGE_EQUALIZED || which == LFDMI_STAGE_ERODED) {
        // equalizeHist output (before the morphology) and the eroded, equalised image of the dim pass: the workspace
        // keeps the un-equalised planes (the monotone LUT commutes with min / max and is applied by their consumers)
        RET(ensure_scratch(ctx, N));
        const uint8_t *plane = (which == LFDMI_STAGE_EQUALIZED ? ctx->gray : ctx->tmp) + slot * N;
        k_apply_lut<<<dim3(512, 1), 256, 0, ctx->stream>>>(plane, ctx->lut + slot * 256, (uint8_t *)ctx->scratch, N);
        KCHK("k_apply_lut");
        src = (const uint8_t *)ctx->scratch;
    } else if (which == LFDMI_STAGE_CANNY || which == LFDMI_STAGE_BOX) {
        const u64 *bits = (which == LFDMI_STAGE_CANNY ? ctx->edgeb : ctx->boxb) + slot * BW;
        RET(ensure_scratch(ctx, N));
        RET(expand_bits(ctx, bits, (uint8_t *)ctx->scratch, 1, h, w));
        src = (const uint8_t *)ctx->scratch;
    } else return fail(ctx, LFDMI_ERR_ARG, "unknown stage");
    RET(out_copy(ctx, dst, 0, src, N, loc));
    return sync(ctx, 1);
}
