/*
 * lfd_oracle.h -- CPU restatement (plain C99) of the lfd.detecttrails per-frame hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under lfd_amd/ may include, link or call this.
 * Allowed users: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.
 *
 * PARITY STATUS: the arithmetic of this path lives in OpenCV (opencv-python, unpinned in the
 * reference's setup.py:27; conda pin 3.4.2 in environment.yml:70), whose sources are NOT under
 * /root/reference and which is not installed in the build image.  The operator semantics below
 * restate OpenCV 3.4.2's published algorithms (imgproc: convertScaleAbs, equalizeHist,
 * morphology, Canny, findContours (Suzuki-Abe), convexHull, rotatingCalipers/minAreaRect,
 * RotatedRect::points, fillPoly, HoughLinesStandard) as summarised in SURVEY.md Appendix A.
 * The reference holds no tests or golden images for them  =>  "parity unpinned" at the OpenCV
 * boundary.  What IS pinned: check_theta / dictify_hough against vectors produced by the
 * reference's own code (tests/golden/tail_fixtures.json), morphology / Sobel / connectivity
 * against scipy.ndimage, hand-derived known answers, analytic Hough bins.
 *
 * Call sites restated (reference file:line):
 *   lfo_prep            lfd/detecttrails/processfield.py:342,346 (bright) / :453-456 (dim),
 *                       lfd/detecttrails/detecttrails.py:124 (flip)
 *   lfo_equalize_hist   processfield.py:347,457
 *   lfo_morph           processfield.py:354 (dilate), :464 (erode), :471 (dilate)
 *   lfo_canny           processfield.py:236
 *   lfo_find_contours   processfield.py:241-246
 *   lfo_min_area_rect, lfo_box_points, lfo_fill_poly, lfo_fit_min_area_rect
 *                       processfield.py:201-263
 *   lfo_hough_lines     processfield.py:370-371,488-489
 *   lfo_check_theta     processfield.py:36-150
 *   lfo_dictify_hough   processfield.py:266-288
 *   lfo_remove_stars    lfd/detecttrails/removestars.py:111-130,212-231
 *   lfo_process_bright  processfield.py:291-388
 *   lfo_process_dim     processfield.py:391-506
 *   lfo_detect_frame    detecttrails.py:119-131
 */
#ifndef LFD_ORACLE_H
#define LFD_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { LFO_U8 = 0, LFO_F32 = 1, LFO_F64 = 2 };
/* prep modes: what is done to the float image before convertScaleAbs */
enum { LFO_PREP_NONE = 0, LFO_PREP_BRIGHT = 1, LFO_PREP_DIM = 2, LFO_PREP_BRIGHT_THEN_DIM = 3 };
enum { LFO_DILATE = 0, LFO_ERODE = 1 };
/* cv2 constants re-exported by detecttrails.py:14-18 */
enum { LFO_RETR_EXTERNAL = 0, LFO_RETR_LIST = 1, LFO_RETR_CCOMP = 2, LFO_RETR_TREE = 3 };
enum { LFO_CHAIN_APPROX_NONE = 1, LFO_CHAIN_APPROX_SIMPLE = 2 };

typedef struct {
    double lwTresh, thetaTresh, lineSetTresh, dro;
    double minAreaRectMinLen;
    double houghMethod;          /* passed as HoughLines' rho (processfield.py:370) */
    int nlinesInSet;
    int contoursMode, contoursMethod;
    int dilate_kh, dilate_kw;    /* kernels: row-major 0/1 masks */
    const uint8_t *dilateKernel;
    int erode_kh, erode_kw;      /* dim only; erodeKernel NULL for bright */
    const uint8_t *erodeKernel;
    double minFlux, addFlux;     /* dim only */
    int gaussKernel;             /* optional smoothing of Canny's input: odd size of a cv2.GaussianBlur-like kernel, 0 = off
                                    (the reference never smooths: cv2.Canny has no Gaussian stage, processfield.py:236) */
    double gaussSigma;           /* <= 0: 0.3*((ksize-1)*0.5 - 1) + 0.8 */
} lfo_params;

typedef struct {
    int32_t status;              /* 0 ok, <0 error (e.g. no Hough line although detection) */
    int32_t found;               /* 0 none, 1 bright, 2 dim */
    float rho, theta;            /* equhough[0][0] */
    int32_t x1, y1, x2, y2;      /* dictify_hough, float32 evaluation */
    int32_t n_lines_equ, n_lines_box;
    int32_t detection;           /* fit_minAreaRect's boolean of the pass that ran last */
    int32_t rejected_by_theta;   /* check_theta returned True in the pass that ran last */
} lfo_result;

typedef struct {
    int defaultxy, maxxy, magcount;
    double pixscale, maxmagdiff;
    double filter_cap;           /* filter_caps[filter] */
    int filter_index;            /* 0..4 = u g r i z */
} lfo_rs_params;

/* A.1 + the numpy masking that precedes it; src may be u8/f32/f64; flip folds cv2.flip(img,0). */
int lfo_prep(const void *src, int dtype, int h, int w, int flip, int mode,
             double minFlux, double addFlux, uint8_t *gray);
/* A.2 */
int lfo_equalize_hist(const uint8_t *src, int h, int w, uint8_t *dst);
int lfo_equalize_lut(const int32_t *hist, int total, uint8_t *lut, int *first_bin, int *constant);
/* A.3 */
int lfo_morph(const uint8_t *src, int h, int w, const uint8_t *kernel, int kh, int kw, int op,
              uint8_t *dst);
/* A.4 */
int lfo_canny(const uint8_t *src, int h, int w, double low_thresh, double high_thresh,
              uint8_t *dst);
int lfo_sobel_mag(const uint8_t *src, int h, int w, int16_t *dx, int16_t *dy, int32_t *mag);
/* Optional Gaussian stage (BASELINE north_star lists one inside Canny; cv2.Canny has none, so it is OFF by default and has
 * no reference call site).  Defined here as cv2.getGaussianKernel(ksize, sigma, CV_32F) applied separably (rows, then
 * columns) in float32 with BORDER_REFLECT_101, accumulation in tap order without FMA, one final round-half-even +
 * saturation.  OpenCV's own 8-bit path uses fixed-point taps since 3.4.1, so this is NOT claimed bit-equal to cv2. */
int lfo_gaussian_kernel(int ksize, double sigma, float *taps);
int lfo_gaussian_blur(const uint8_t *src, int h, int w, int ksize, double sigma, uint8_t *dst);
int lfo_fit_min_area_rect_g(const uint8_t *img, int h, int w, int contoursMode, int contoursMethod,
                            double minAreaRectMinLen, double lwTresh, int gaussKernel, double gaussSigma,
                            uint8_t *box_img, int32_t *detection, int32_t *n_boxes);
/* A.5: returns contours as a flat (x,y) int32 list + offsets[n+1]; caller frees with lfo_free */
int lfo_find_contours(const uint8_t *img, int h, int w, int mode, int32_t **points,
                      int32_t **offsets, int32_t **is_hole, int32_t *n_contours);
void lfo_free(void *p);
/* A.6: rect = cx, cy, w, h, angle_deg (float32) */
int lfo_convex_hull(const int32_t *pts, int n, int32_t *hull /* 2*n ints */, int *n_hull);
int lfo_min_area_rect(const int32_t *pts, int n, float rect[5]);
void lfo_box_points(const float rect[5], float box[8]);
void lfo_debug_trig(int n, const double *y, const double *x, float *angle_deg, float *cos_half, float *sin_half);
int lfo_fill_poly(uint8_t *img, int h, int w, const int32_t *pts, int npts, uint8_t color);
int lfo_fit_min_area_rect(const uint8_t *img, int h, int w, int contoursMode, int contoursMethod,
                          double minAreaRectMinLen, double lwTresh, uint8_t *box_img,
                          int32_t *detection, int32_t *n_boxes);
/* A.7: lines = (rho,theta) float32 pairs sorted by votes desc; n_lines = total found */
int lfo_hough_lines(const uint8_t *img, int h, int w, double rho, double theta, int threshold,
                    int max_lines, float *lines, int32_t *n_lines);
int lfo_hough_accum(const uint8_t *img, int h, int w, double rho, double theta,
                    int32_t *accum /* (numangle+2)*(numrho+2) */, int *numangle, int *numrho);
void lfo_hough_dims(int h, int w, double rho, double theta, int *numangle, int *numrho);
/* returns 1 = True (reject), 0 = None (accept) */
int lfo_check_theta(const float *h1, int n1, const float *h2, int n2, int navg, double dro,
                    double thetaTresh, double lineSetTresh);
void lfo_dictify_hough(int n_x, int n_y, float rho, float theta, int32_t out[4]);
/* removestars.py: catalogue columns as float32 [n][5] and int32 [n]; img f32 (h,w) in place */
int lfo_remove_stars(float *img, int h, int w, int n_obj, const float *rowc, const float *colc,
                     const float *psfmag, const float *petro90, const int32_t *nobserve,
                     const int32_t *ndetect, const lfo_rs_params *p);

int lfo_process_bright(const void *img, int dtype, int h, int w, int flip, const lfo_params *p,
                       lfo_result *res, uint8_t *equ_out, uint8_t *box_out);
int lfo_process_dim(const void *img, int dtype, int h, int w, int flip, int after_bright,
                    const lfo_params *p, lfo_result *res, uint8_t *equ_out, uint8_t *box_out);
/* removestars (optional) -> flip -> bright -> dim, on a float32 frame (mutated like the reference) */
int lfo_detect_frame(float *img, int h, int w, const lfo_params *bright, const lfo_params *dim,
                     int n_obj, const float *rowc, const float *colc, const float *psfmag,
                     const float *petro90, const int32_t *nobserve, const int32_t *ndetect,
                     const lfo_rs_params *rs, lfo_result *res);

#ifdef __cplusplus
}
#endif
#endif
