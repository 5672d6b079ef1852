/*
 * lfdmi.h -- C-ABI of liblfdmi.so: the MI355X (gfx950) implementation of the per-frame hot
 * path of lfd.detecttrails.  Plain pointers and sizes only; loaded with ctypes.CDLL by
 * lfd_amd/_native.py (see INTEGRATION.md for the binding a maintainer of the reference adds).
 *
 * The reference has no FFI of its own: its seam is the Python call boundary of
 * lfd/detecttrails/__init__.py:69-71.  Each entry point below names the reference interface
 * it stands in for (file:line under /root/reference).
 *
 * Conventions
 *   - every function returns 0 on success, a negative lfdmi_status otherwise;
 *     lfdmi_last_error(ctx) gives the text.  HIP errors never abort the process.
 *   - images are row-major, C-contiguous, `n` images of h x w back to back.
 *   - `loc` says where caller buffers live: LFDMI_HOST (the library stages them) or
 *     LFDMI_DEVICE (hipMalloc'd / torch CUDA memory on ctx's device, used in place).
 *   - a frame must fit the ctx in BOTH dimensions (h <= max_h and w <= max_w), not only in area.
 *   - a ctx is bound to one device, is not thread-safe, and runs everything on one HIP
 *     stream (its own, or the caller's via lfdmi_set_stream).  Calls return after the
 *     stream has drained (synchronous at the ABI).
 *   - the library never retains or frees caller pointers.
 */
#ifndef LFDMI_H
#define LFDMI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LFDMI_VERSION 300

enum lfdmi_status {
    LFDMI_OK = 0,
    LFDMI_ERR_ARG = -1,         /* bad argument (shape, dtype, NULL) */
    LFDMI_ERR_DTYPE = -2,       /* dtype not valid for this operation (dim pass on uint8) */
    LFDMI_ERR_HIP = -3,         /* HIP runtime error; see lfdmi_last_error */
    LFDMI_ERR_UNSUPPORTED = -4, /* knob value not implemented (e.g. CHAIN_APPROX_TC89_*) */
    LFDMI_ERR_CAPACITY = -5,    /* frame larger than the ctx was created for / workspace overflow */
    LFDMI_ERR_NOLINES = -6      /* per-frame status: HoughLines returned no line although
                                   fit_minAreaRect detected (reference: TypeError, logged) */
};

/* LFDMI_HOST_PINNED (lfdmi_detect_batch / _raw only): host memory obtained from lfdmi_host_alloc -- the DMA engines read it in
 * place, without the staging copy ordinary host frames take */
enum { LFDMI_HOST = 0, LFDMI_DEVICE = 1, LFDMI_HOST_PINNED = 2 };
/* LFDMI_F32_BE (lfdmi_detect_batch_raw only; LFDMI_DEVICE frames of that type are byte-swapped IN PLACE): big-endian float32, the raw data unit of a BITPIX = -32 FITS image (what
 * fitsio hands the reference after its own byte swap, detecttrails.py:113); swapped on the device after the upload */
enum { LFDMI_U8 = 0, LFDMI_F32 = 1, LFDMI_F64 = 2, LFDMI_F32_BE = 3 };
/* numpy masking done before cv2.convertScaleAbs */
enum { LFDMI_PREP_NONE = 0, LFDMI_PREP_BRIGHT = 1, LFDMI_PREP_DIM = 2, LFDMI_PREP_BRIGHT_THEN_DIM = 3 };
/* cv2 constants re-exported by lfd/detecttrails/detecttrails.py:14-18 */
enum { LFDMI_RETR_EXTERNAL = 0, LFDMI_RETR_LIST = 1, LFDMI_RETR_CCOMP = 2, LFDMI_RETR_TREE = 3 };
enum { LFDMI_CHAIN_APPROX_NONE = 1, LFDMI_CHAIN_APPROX_SIMPLE = 2, LFDMI_CHAIN_APPROX_TC89_L1 = 3,
       LFDMI_CHAIN_APPROX_TC89_KCOS = 4 };
/* lfdmi_get_stage selectors (device images of the last process/detect call, per slot) */
/* GRAY: convertScaleAbs output; EQUALIZED: equalizeHist(gray) (1equBRIGHT / 6equDIM); ERODED: erode(equalized)
 * (7erodedDIM); EQU: the dilated image Canny and HoughLines see (2dilateBRIGHT / 8openedDIM); CANNY: the edge map;
 * BOX: box_img (3contoursBRIGHT / 9contoursDIM) -- debug dump names of processfield.py:349-378, :459-496 */
enum { LFDMI_STAGE_GRAY = 0, LFDMI_STAGE_EQU = 1, LFDMI_STAGE_CANNY = 2, LFDMI_STAGE_BOX = 3, LFDMI_STAGE_ERODED = 4,
       LFDMI_STAGE_EQUALIZED = 5 };

typedef struct lfdmi_ctx lfdmi_ctx;

/* The keys of params_bright / params_dim (detecttrails.py:202-230); key names == argument
 * names of process_field_bright/dim (processfield.py:291-293, :391-394).  Kernels are
 * row-major 0/1 masks in HOST memory. */
typedef struct {
    double lwTresh, thetaTresh, lineSetTresh, dro;
    double minAreaRectMinLen;
    double houghMethod;           /* passed to HoughLines as rho (processfield.py:370,488) */
    int32_t nlinesInSet;          /* 1..LFDMI_MAX_SET_LINES */
    int32_t contoursMode, contoursMethod;
    int32_t dilate_kh, dilate_kw;
    const uint8_t *dilateKernel;
    int32_t erode_kh, erode_kw;   /* dim only */
    const uint8_t *erodeKernel;
    double minFlux, addFlux;      /* dim only */
    /* Optional smoothing of Canny's input.  BASELINE's north_star lists a Gaussian stage inside Canny; cv2.Canny
     * (processfield.py:236) has none, so it is OFF unless gaussKernel > 0 and has no reference call site.  Odd size
     * 1..31; gaussSigma <= 0: cv2.getGaussianKernel's default for that size.  Semantics: lfdmi_gaussian_blur. */
    int32_t gaussKernel;
    double gaussSigma;
} lfdmi_params;

#define LFDMI_MAX_SET_LINES 64
#define LFDMI_MAX_MORPH_K 31

/* params_removestars (detecttrails.py:231-239) for one filter */
typedef struct {
    int32_t defaultxy, maxxy, magcount;
    double pixscale, maxmagdiff;
    double filter_cap;            /* filter_caps[filter] */
    int32_t filter_index;         /* 0..4 = u g r i z */
} lfdmi_rs_params;

/* photoObj columns read by removestars.py:96-104, padded to max_obj rows per frame */
typedef struct {
    int32_t max_obj;
    const int32_t *count;         /* [n] objects per frame */
    const float *rowc, *colc, *psfmag, *petro90; /* [n][max_obj][5] */
    const int32_t *nobserve, *ndetect;           /* [n][max_obj] */
    int32_t loc;                  /* where these arrays live */
} lfdmi_catalog;

/* one record per frame; what process_field needs to write a results row */
typedef struct {
    int32_t status;               /* 0 or a negative lfdmi_status for this frame */
    int32_t found;                /* 0 none, 1 bright pass, 2 dim pass */
    float rho, theta;             /* equhough[0][0] (processfield.py:384,502) */
    int32_t x1, y1, x2, y2;       /* dictify_hough, float32 evaluation (processfield.py:266-288) */
    int32_t n_lines_equ, n_lines_box;
    int32_t detection;            /* fit_minAreaRect's flag in the last pass that ran */
    int32_t rejected_by_theta;    /* check_theta returned True in the last pass that ran */
} lfdmi_result;

/* Workspace sizing.  Per in-flight frame the workspace holds dense 8-bit planes, bit rows and a set of
 * tables whose theoretical maxima (a checkerboard: H*(W/2+1) runs, N/2 contours, 2N contour rows, N Hough
 * chunks) are ~100 B/px, while sky frames use less than 1 % of that (tools/cap_survey.py).  A context is
 * therefore created with the capacities below (per frame; N = max_h*max_w); a frame that needs more in
 * any table is detected on the device (no table is ever indexed past its capacity), and the library
 * runs it again, alone, through a worst-case workspace it keeps for that purpose (created on first
 * use, one frame in flight), so no input can fail for lack of table space.  A field <= 0 means "the
 * theoretical maximum"; caps == NULL means the defaults of lfdmi_default_caps. */
typedef struct {
    int32_t run_cap;   /* runs per bit image (Canny candidates / background of the edge image); default N/16 */
    int32_t key_cap;   /* contours (edge components + holes); default N/256 */
    int32_t slot_cap;  /* contour rows (one (xmin,xmax) slot per row of every contour); default N/16 */
    int32_t list_cap;  /* Hough input chunks (pieces of pixel runs, <= 64 px) per image and list; default N/16 */
    int32_t peak_cap;  /* Hough local maxima per image (rounded up to a power of two); default 65536 */
    double min_rho;    /* HoughLines accumulators are sized for rho >= min_rho (theta >= pi/180); default 5;
                          a call with a finer rho runs through the worst-case workspace */
} lfdmi_caps;
void lfdmi_default_caps(int max_h, int max_w, lfdmi_caps *out);

int lfdmi_version(void);
/* max_inflight = frames processed concurrently (workspace is sized for that many, default capacities). */
int lfdmi_ctx_create(int device, int max_h, int max_w, int max_inflight, lfdmi_ctx **out);
int lfdmi_ctx_create_sized(int device, int max_h, int max_w, int max_inflight, const lfdmi_caps *caps,
                           lfdmi_ctx **out);
/* device bytes the workspace holds; frames re-run through the worst-case workspace since creation */
int64_t lfdmi_ctx_bytes(lfdmi_ctx *ctx);
int64_t lfdmi_spill_count(lfdmi_ctx *ctx);
/* What a context has had to do besides the fast path since it was created (out[0 .. n), n <= LFDMI_STAT_COUNT): none of
 * these changes a result, all of them cost time, so a caller (bench.py prints them) can tell a slow run from a busy one. */
enum {
    LFDMI_STAT_SPILLED = 0,        /* frames run again alone in the worst-case workspace (= lfdmi_spill_count) */
    LFDMI_STAT_SCAN_GIVEUPS = 1,   /* chunks in which the one-launch run scan gave up its look-back (GPU shared with other work);
                                      each switches the context to the three-launch scan for a while and reruns that chunk */
    LFDMI_STAT_GENERAL_RERUNS = 2, /* chunks run again because a frame needed the multi-workgroup run kernels while they were off */
    LFDMI_STAT_GENERAL_CHUNKS = 3, /* chunks that ran with the multi-workgroup run kernels switched on */
    LFDMI_STAT_CHUNKS = 4,         /* chunks processed by lfdmi_detect_batch / the per-pass entry points */
    LFDMI_STAT_CAP_GROWTHS = 5,    /* times the per-frame tables were enlarged after frames overflowed them */
    LFDMI_STAT_SCAN_FUSED_ON = 6,  /* 1 while the one-launch run scan is in use */
    LFDMI_STAT_COUNT = 7
};
int lfdmi_get_stats(lfdmi_ctx *ctx, int64_t *out, int n);
void lfdmi_ctx_destroy(lfdmi_ctx *ctx);
const char *lfdmi_last_error(lfdmi_ctx *ctx);
/* run on the caller's hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); NULL = own */
int lfdmi_set_stream(lfdmi_ctx *ctx, void *hip_stream);
int lfdmi_max_inflight(lfdmi_ctx *ctx);

/* ---- per-operator entry points (single images from processfield.py, and parity tests) ---- */

/* img[img<0]=0 / img[img<minFlux]=0; img[img>0]+=addFlux, then cv2.convertScaleAbs
 * (processfield.py:342,346 / :453-456), optionally after cv2.flip(img,0)
 * (detecttrails.py:124).  hist (n x 256 int32) may be NULL. */
int lfdmi_prep_u8(lfdmi_ctx *ctx, const void *src, int dtype, int n, int h, int w, int flip,
                  int mode, double minFlux, double addFlux, uint8_t *gray, int32_t *hist, int loc);
/* cv2.equalizeHist (processfield.py:347,457) */
int lfdmi_equalize_hist(lfdmi_ctx *ctx, const uint8_t *src, int n, int h, int w, uint8_t *dst,
                        int loc);
/* cv2.dilate / cv2.erode(img, kernel) (processfield.py:354,464,471); kernel on the host */
int lfdmi_dilate(lfdmi_ctx *ctx, const uint8_t *src, int n, int h, int w, const uint8_t *kernel,
                 int kh, int kw, uint8_t *dst, int loc);
int lfdmi_erode(lfdmi_ctx *ctx, const uint8_t *src, int n, int h, int w, const uint8_t *kernel,
                int kh, int kw, uint8_t *dst, int loc);
/* Optional Gaussian stage (no reference call site, see lfdmi_params.gaussKernel): cv2.getGaussianKernel(ksize, sigma,
 * CV_32F) applied separably in float32 (rows, then columns; BORDER_REFLECT_101; taps accumulated in order without FMA),
 * one final round-half-even + saturation.  OpenCV's own 8-bit path uses fixed-point taps since 3.4.1, so this is the
 * build's definition (oracle: lfo_gaussian_blur), not a cv2 parity claim. */
int lfdmi_gaussian_blur(lfdmi_ctx *ctx, const uint8_t *src, int n, int h, int w, int ksize, double sigma,
                        uint8_t *dst, int loc);
/* cv2.Canny(img, low, high) with aperture 3, L1 gradient (processfield.py:236) */
int lfdmi_canny(lfdmi_ctx *ctx, const uint8_t *src, int n, int h, int w, double low, double high,
                uint8_t *dst, int loc);
/* fit_minAreaRect (processfield.py:201-263): Canny(0,255) -> contours -> minAreaRect ->
 * side/elongation filter -> boxPoints -> int32 -> fillPoly.  box_img n*h*w u8 (may be NULL),
 * detection / n_boxes n x int32 (may be NULL). */
int lfdmi_fit_min_area_rect(lfdmi_ctx *ctx, const uint8_t *img, int n, int h, int w,
                            int contoursMode, int contoursMethod, double minAreaRectMinLen,
                            double lwTresh, uint8_t *box_img, int32_t *detection,
                            int32_t *n_boxes, int loc);
/* cv2.HoughLines(img, rho, theta, threshold) (processfield.py:370-371,488-489).
 * lines: n x max_lines x 2 float32 (rho, theta), sorted by votes descending;
 * n_lines: total number of lines found per image (may exceed max_lines). */
int lfdmi_hough_lines(lfdmi_ctx *ctx, const uint8_t *img, int n, int h, int w, double rho,
                      double theta, int threshold, int max_lines, float *lines, int32_t *n_lines,
                      int loc);
/* the raw (numangle+2) x (numrho+2) int32 vote accumulator of the same call */
int lfdmi_hough_accum(lfdmi_ctx *ctx, const uint8_t *img, int n, int h, int w, double rho,
                      double theta, int32_t *accum, int loc);
void lfdmi_hough_dims(int h, int w, double rho, double theta, int *numangle, int *numrho);
/* remove_stars (removestars.py:212-231) on float32 frames, in place */
int lfdmi_remove_stars(lfdmi_ctx *ctx, float *img, int n, int h, int w, const lfdmi_catalog *cat,
                       const lfdmi_rs_params *rs, int loc);

/* ---- whole passes ---- */

/* process_field_bright (processfield.py:291-388) on n images.  lines_equ / lines_box
 * (n x nlinesInSet x 2 float32, may be NULL) receive the first nlinesInSet Hough lines of
 * each set so the caller can run check_theta itself; results always filled. */
int lfdmi_process_bright(lfdmi_ctx *ctx, const void *img, int dtype, int n, int h, int w, int flip,
                         const lfdmi_params *p, lfdmi_result *results, float *lines_equ,
                         float *lines_box, int loc);
/* process_field_dim (processfield.py:391-506); after_bright = the array was already clamped
 * by the bright pass (detecttrails.py:125,129 share one array) */
int lfdmi_process_dim(lfdmi_ctx *ctx, const void *img, int dtype, int n, int h, int w, int flip,
                      int after_bright, const lfdmi_params *p, lfdmi_result *results,
                      float *lines_equ, float *lines_box, int loc);
/* One pass with HoughLines evaluated at several rho ("multi-scale Hough", BASELINE.json configs[4]; the
 * reference itself always calls the classic transform once, processfield.py:370-371,488-489): the front
 * end (mask .. fit_minAreaRect) runs once, then HoughLines(equ) / HoughLines(box_img) / check_theta for
 * every rhos[s].  results[s * n + i] is exactly what lfdmi_process_bright / _dim returns for frame i with
 * houghMethod = rhos[s] (p->houghMethod itself is ignored).  dim != 0: process_field_dim (after_bright as
 * in lfdmi_process_dim); dim == 0: process_field_bright.  1 <= n_scales <= LFDMI_MAX_SCALES. */
#define LFDMI_MAX_SCALES 4
int lfdmi_process_multiscale(lfdmi_ctx *ctx, const void *img, int dtype, int n, int h, int w, int flip,
                             int dim, int after_bright, const lfdmi_params *p, int n_scales,
                             const double *rhos, lfdmi_result *results, int loc);
/* process_field's hot part (detecttrails.py:119-131) for n float32 frames:
 * remove_stars (cat may be NULL) -> flip -> bright -> dim where bright found nothing.
 * frames are mutated by remove_stars only, as in the reference.  results: n records in HOST
 * memory. */
int lfdmi_detect_batch(lfdmi_ctx *ctx, float *frames, int n, int h, int w,
                       const lfdmi_catalog *cat, const lfdmi_rs_params *rs,
                       const lfdmi_params *bright, const lfdmi_params *dim, lfdmi_result *results,
                       int loc);
/* lfdmi_detect_batch with the frames' element type given: LFDMI_F32, or LFDMI_F32_BE for HOST / HOST_PINNED frames holding the
 * big-endian data unit of a FITS image as read from the file (DetectTrails.process reads frame files straight into pinned
 * memory and leaves the byte swap to the device: detecttrails.py:73-117 is a read + swap + copy per frame in the reference).
 * Big-endian frames are treated as a read-only input: remove_stars' squares are applied inside the library (masked as
 * the bright sweep loads the values; a frame that has to be run again alone takes its own catalogue entry) and the caller's
 * bytes -- a file's data unit -- stay as they are; LFDMI_F32 frames are blotted in place as in lfdmi_detect_batch (complete
 * when the call returns: for device-resident frames the zero fill runs on a side stream during the call).  LFDMI_F32_BE frames
 * in DEVICE memory (data units decompressed there: lfdmi_bz2_frames) are working memory of the caller's, not a file's bytes: they are
 * byte-swapped in place and then treated like any LFDMI_F32 device frames. */
int lfdmi_detect_batch_raw(lfdmi_ctx *ctx, void *frames, int dtype, int n, int h, int w,
                           const lfdmi_catalog *cat, const lfdmi_rs_params *rs,
                           const lfdmi_params *bright, const lfdmi_params *dim, lfdmi_result *results,
                           int loc);
/* page-locked host memory for LFDMI_HOST_PINNED frames, placed on the NUMA node next to ctx's GPU (the allocating thread
 * is bound to the GPU's local CPUs; LFDMI_NUMA_PIN=0 in the environment disables the binding).  Free with lfdmi_host_free
 * (any live ctx of the same device, or NULL). */
int lfdmi_host_alloc(lfdmi_ctx *ctx, uint64_t bytes, void **out);
int lfdmi_host_free(lfdmi_ctx *ctx, void *p);
/* ---- FITS ingest on host threads (detecttrails.py:73-117 reads a frame with fitsio.read, removestars.py:96-104 the photoObj
 * columns) -- no GPU involved; plain files only (a .fits.bz2 is decompressed by the caller) ----
 * lfdmi_fits_read_frames: the primary-HDU data units of n files into dst (n x h x w big-endian float32 slots, e.g. memory
 * from lfdmi_host_alloc) with `threads` reader threads.  status[i]: 0 = BITPIX -32, NAXIS 2, h x w, no BSCALE / BZERO: the data
 * unit is in slot i as it is in the file; 1 = a FITS file of another kind (slot untouched: use a general reader); -1 = cannot
 * be opened; -2 = truncated / malformed.  hdr (may be NULL): the raw header of file i (80-byte cards up to its padded END
 * block) is copied to hdr + i * hdr_cap, hdr_len[i] = its full length (> hdr_cap: the copy is cut). */
int lfdmi_fits_read_frames(const char *const *paths, int n, int h, int w, void *dst, int threads, int32_t *status, char *hdr,
                           int hdr_cap, int32_t *hdr_len);
/* lfdmi_fits_read_photoobj: ROWC, COLC, PSFMAG, PETROTH90 (float32[5] per object) and NOBSERVE, NDETECT (integers) of the
 * binary table in HDU 1 of n files into the padded lfdmi_catalog arrays in host memory ([n][max_obj][5] / [n][max_obj]) and
 * count[n].  status[i]: 0 ok; 2 / 3 = ok but a float column holds a NaN / an infinity (math.ceil raises on those in
 * removestars.py:113-130: the caller makes it that frame's error); 1 = declined (more than max_obj rows, scaled columns,
 * another column layout: use a general reader); -1 cannot be opened; -2 malformed; -3 a wanted column is missing. */
int lfdmi_fits_read_photoobj(const char *const *paths, int n, int max_obj, float *rowc, float *colc, float *psfmag,
                             float *petro90, int32_t *nobserve, int32_t *ndetect, int32_t *count, int threads,
                             int32_t *status);
/* lfdmi_bz2_find_blocks: bit offsets of the block magics and the end-of-stream magic of a bzip2 file in host memory
 * (out[i] = bit offset * 2 + 1 for the end-of-stream magic; returns their number, the first `cap` stored) -- the blocks of a
 * .fits.bz2 frame are then decoded side by side (the reference pipes the file through bunzip2: detecttrails.py:81-109). */
int64_t lfdmi_bz2_find_blocks(const uint8_t *data, uint64_t n, uint64_t *out, int64_t cap);
/* ---- bzip2 on the device --------------------------------------------------------------------------------------------------
 * SDSS serves frames as frame-*.fits.bz2; the reference decompresses every one before it reads it (detecttrails.py:81-109:
 * `bunzip2` into $FITS_DUMP, then fitsio.read) at ~0.4 s per frame and core.  lfdmi_bz2_decode_batch decompresses n whole files
 * at once on the GPU (a frame is ~14 independent 900 kB blocks: Huffman / move-to-front a wave per block, inverse
 * Burrows-Wheeler transform as a list ranking, run-length layer as a scan; every block's CRC and the stream's CRC are
 * checked).  A handle owns its own stream and buffers and is independent of any lfdmi_ctx (one thread per handle).
 *   src + src_off[i], src_len[i]  file i's bytes in host memory (ordinary or page-locked);
 *   out_cap                        room per decompressed file;
 *   head, head_bytes               if head_bytes > 0: the first head_bytes of every decompressed file, side by side, in host
 *                                  memory (for parsing FITS headers; zero-filled beyond a file's end);
 *   out_len[i], status[i]          decompressed size, and 0 = decoded and checked, or why not (LFDMI_BZ2_*: the caller then
 *                                  decompresses that file on the host, which also produces the reference's error for broken
 *                                  files).  Concatenated streams (`bzip2 -c a b`, pbzip2) are one file.  Trailing bytes, randomised
 *                                  blocks of bzip2 < 0.9.5 and anything else unexpected: declined, not guessed at.
 * The decompressed files stay on the device until the handle's next lfdmi_bz2_decode_batch; lfdmi_bz2_fetch /
 * lfdmi_bz2_fetch_many copy ranges of them to host (loc LFDMI_HOST / LFDMI_HOST_PINNED) or device (LFDMI_DEVICE) memory. */
typedef struct lfdmi_bz2 lfdmi_bz2;
enum { LFDMI_BZ2_OK = 0, LFDMI_BZ2_MAGIC = 1, LFDMI_BZ2_RANDOMISED = 2, LFDMI_BZ2_HEADER = 3, LFDMI_BZ2_DATA = 4, LFDMI_BZ2_LENGTH = 5,
       LFDMI_BZ2_ORIGPTR = 6, LFDMI_BZ2_CRC = 7, LFDMI_BZ2_CYCLE = 8, LFDMI_BZ2_SIZE = 9, LFDMI_BZ2_STREAM = 10 };
int lfdmi_bz2_create(int device, lfdmi_bz2 **out);
void lfdmi_bz2_destroy(lfdmi_bz2 *z);
const char *lfdmi_bz2_last_error(lfdmi_bz2 *z);
int lfdmi_bz2_decode_batch(lfdmi_bz2 *z, const void *src, const uint64_t *src_off, const uint64_t *src_len, int n, uint64_t out_cap,
                           void *head, uint64_t head_bytes, uint64_t *out_len, int32_t *status);
int lfdmi_bz2_fetch(lfdmi_bz2 *z, int i, uint64_t off, uint64_t nbytes, void *dst, int loc);
int lfdmi_bz2_fetch_many(lfdmi_bz2 *z, int n, const int32_t *file, const uint64_t *off, const uint64_t *nbytes, void *const *dst, int loc);
/* Device memory of the handle's own (two buffers, which = 0 / 1: one chunk is decoded while the previous one is processed) to
 * gather decoded data units into -- lfdmi_bz2_fetch_many(..., LFDMI_DEVICE) -- and to hand to lfdmi_detect_batch_raw(...,
 * LFDMI_F32_BE, ..., LFDMI_DEVICE): compressed frames then cross PCIe once, compressed, and never return to the host. */
int lfdmi_bz2_frames(lfdmi_bz2 *z, int which, uint64_t bytes, void **dev);
/* optional: allocate now what a batch of n_files files / n_blocks blocks (out_cap bytes of output each, compressed_bytes in all) will
 * need, instead of inside the first lfdmi_bz2_decode_batch (tens of GB for a chunk of frames; can run beside the first reads) */
int lfdmi_bz2_reserve(lfdmi_bz2 *z, int n_files, int64_t n_blocks, uint64_t out_cap, uint64_t compressed_bytes);
/* milliseconds of the last batch: upload + magic search, Huffman / move-to-front, sort, inverse BWT walks, run-length + CRC + output */
int lfdmi_bz2_timings(lfdmi_bz2 *z, float *ms5);
/* Which calls keep the 8-bit stage images (gray, eroded, equalised+dilated: what the reference's debug PNGs show) for
 * lfdmi_get_stage.  mode -1 (default): the per-pass entry points (lfdmi_process_bright / _dim / _multiscale) do,
 * lfdmi_detect_batch does not; 0: no call does (batches through the per-pass entry points: an image per frame less to
 * write, and the dim pass may fuse its front end); 1: every call does.  The edge map and box image are always available. */
int lfdmi_set_stage_images(lfdmi_ctx *ctx, int mode);
/* copy a stage image (u8, h x w) of in-flight slot `slot` of the LAST call to dst; h, w must be the shape of
 * that call (LFDMI_ERR_ARG otherwise: dst is then too small or too large for what the workspace holds), and for the
 * 8-bit images the call must have kept them (lfdmi_set_stage_images; LFDMI_ERR_ARG otherwise) */
int lfdmi_get_stage(lfdmi_ctx *ctx, int slot, int which, int h, int w, uint8_t *dst, int loc);
/* diagnostics: the work counters of in-flight slots [slot0, slot0 + n) as left by the LAST pass
 * (LFDMI_COUNTERS int32 values per slot; order: keys, row slots, rectangles, equ list entries, box
 * list entries (pixel chunks the Hough kernels vote with), equ peaks, box peaks, overflow flag,
 * detection, tall keys, candidate words, background words, candidate runs, background runs, medium
 * keys, non-zero pixels of equ, of box_img, active 64 x 16 tiles, entries of the second (longer-chunk) Hough lists of equ / box) */
#define LFDMI_COUNTERS 20
int lfdmi_get_counters(lfdmi_ctx *ctx, int slot0, int n, int32_t *dst);
/* per-kernel timing for bench.py's roofline entry: when enabled, every kernel launch is
 * bracketed by HIP events on the launch stream; lfdmi_get_timing returns, per timing slot,
 * the summed device time (ms), the number of launches and the number of frames those launches
 * actually worked on (a dim-pass launch only works on frames the bright pass left undecided,
 * a Hough launch only on frames with a detected rectangle) since lfdmi_enable_timing(ctx, 1).
 * lfdmi_timing_slots() slots, named by lfdmi_timing_name(i) (the kernel's name).
 * lfdmi_timing_select(ctx, mask) restricts the bracketing to the slots whose bit is set in mask
 * (0: all again): the ~80 event records of a fully timed step cost ~6 % of it, a handful per step
 * do not, so bench.py times only the few largest kernels inside its timed region. */
int lfdmi_enable_timing(lfdmi_ctx *ctx, int on);
int lfdmi_timing_select(lfdmi_ctx *ctx, uint64_t mask);
int lfdmi_get_timing(lfdmi_ctx *ctx, float *ms, int32_t *launches, int64_t *units);
int lfdmi_timing_slots(void);
const char *lfdmi_timing_name(int slot);
/* developer tool (LFDMI_FRAME_PROFILE=1 in the environment when the context is created): per-phase clocks
 * (8 x int64 per slot, 10 ns ticks) of the last per-frame contour kernel launch, slots 0 .. n-1 */
int lfdmi_debug_frame_profile(lfdmi_ctx *ctx, int n, long long *dst);
/* test entry point: the tail of a pass on line sets handed in from outside -- the device's check_theta (k_finalize;
 * reference: lfd/detecttrails/processfield.py:36-150, zero fill :89-102) and the library's host-side dictify_hough
 * (processfield.py:266-288), exactly as lfdmi_detect_batch runs them on its own Hough lines.  h1 / h2: [n][kmax][2] float32
 * (rho, theta), n1 / n2: lines per set; out[i].rejected_by_theta = check_theta's True, out[i].found = which (1 / 2) when it
 * returns None, with rho / theta / x1 .. y2 filled in.  tests/test_gpu_stages.py feeds it the reference-generated vectors of
 * tests/golden/tail_fixtures.json. */
int lfdmi_debug_tail(lfdmi_ctx *ctx, int n, int kmax, const float *h1, const int32_t *n1, const float *h2, const int32_t *n2,
                     int navg, double dro, double thetaTresh, double lineSetTresh, int which, int h, int w, lfdmi_result *out);
/* developer hook for tests of the error path: the NEXT lfdmi_detect_batch call on this context returns LFDMI_ERR_ARG at the
 * top of its chunk number `chunk` (0-based; a chunk is the feed's unit for host frames, max_inflight frames otherwise),
 * after the earlier chunks ran normally; -1 disarms.  The context stays usable. */
int lfdmi_debug_fail_chunk(lfdmi_ctx *ctx, int chunk);
/* developer check: the device's float32 results of the three libm calls on minAreaRect's accept / reject path (angle in
 * degrees of atan2(y, x) as cv::minAreaRect rounds it; cos / sin of that angle times 0.5 as RotatedRect::points does), for
 * n host operand pairs -- compared with the host libm by tests/test_gpu_stages.py */
int lfdmi_debug_trig(lfdmi_ctx *ctx, int n, const double *y, const double *x, float *angle_deg, float *cos_half, float *sin_half);

#ifdef __cplusplus
}
#endif
#endif
