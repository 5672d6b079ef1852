/*
 * lfd_oracle.c -- CPU restatement of the lfd.detecttrails hot path.  See lfd_oracle.h.
 *
 * TEST INFRASTRUCTURE ONLY (checker for tests/, smoke(), bench.py cpu_baseline).
 * PARITY UNPINNED at the OpenCV boundary (no cv2 in the image, no golden data in the
 * reference); semantics follow OpenCV 3.4.2 as summarised in SURVEY.md Appendix A.
 *
 * Build: gcc -O2 -std=c99 -ffp-contract=off -fno-fast-math -fPIC -shared (see Makefile).
 * -ffp-contract=off matters: OpenCV's baseline x86-64 build evaluates float expressions
 * without FMA, and the vote bin of HoughLines / the corners of boxPoints depend on it.
 */
#include "lfd_oracle.h"

#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define LFO_PI 3.1415926535897932384626433832795

void lfo_free(void *p) { free(p); }

/* ------------------------------------------------------------------------------------------
 * A.1  convertScaleAbs (alpha 1, beta 0) preceded by the numpy masking of the caller.
 *   bright: processfield.py:342   img[img < 0] = 0
 *   dim:    processfield.py:453-454  img[img < minFlux] = 0 ; img[img > 0] += addFlux
 *   saturate_cast<uchar>(|x|): cvRound = round-half-to-even, clamp to 0..255, NaN -> 0.
 * ------------------------------------------------------------------------------------------ */
static uint8_t sat_u8_from_f32(float v) {
    float a = fabsf(v);
    if (a != a) return 0;          /* NaN: cvtps2dq gives INT_MIN, packs saturate to 0 */
    if (a >= 255.5f) return 255;
    return (uint8_t)lrintf(a);     /* default rounding mode = nearest-even */
}

static uint8_t sat_u8_from_f64(double v) {
    double a = fabs(v);
    if (a != a) return 0;
    if (a >= 255.5) return 255;
    return (uint8_t)lrint(a);
}

int lfo_prep(const void *src, int dtype, int h, int w, int flip, int mode, double minFlux,
             double addFlux, uint8_t *gray) {
    if (h <= 0 || w <= 0) return -1;
    if (dtype == LFO_U8 && (mode == LFO_PREP_DIM || mode == LFO_PREP_BRIGHT_THEN_DIM))
        return -2; /* numpy refuses `uint8 += float` (same_kind casting): a frame error */
    for (int r = 0; r < h; r++) {
        int sr = flip ? (h - 1 - r) : r; /* cv2.flip(img, 0): dst[r] = src[H-1-r] */
        uint8_t *d = gray + (size_t)r * w;
        if (dtype == LFO_U8) {
            memcpy(d, (const uint8_t *)src + (size_t)sr * w, (size_t)w);
        } else if (dtype == LFO_F32) {
            const float *s = (const float *)src + (size_t)sr * w;
            float mf = (float)minFlux, af = (float)addFlux;
            for (int c = 0; c < w; c++) {
                float x = s[c];
                if (mode == LFO_PREP_BRIGHT || mode == LFO_PREP_BRIGHT_THEN_DIM)
                    if (x < 0.0f) x = 0.0f;
                if (mode == LFO_PREP_DIM || mode == LFO_PREP_BRIGHT_THEN_DIM) {
                    if (x < mf) x = 0.0f;
                    if (x > 0.0f) x = x + af;
                }
                d[c] = sat_u8_from_f32(x);
            }
        } else if (dtype == LFO_F64) {
            const double *s = (const double *)src + (size_t)sr * w;
            for (int c = 0; c < w; c++) {
                double x = s[c];
                if (mode == LFO_PREP_BRIGHT || mode == LFO_PREP_BRIGHT_THEN_DIM)
                    if (x < 0.0) x = 0.0;
                if (mode == LFO_PREP_DIM || mode == LFO_PREP_BRIGHT_THEN_DIM) {
                    if (x < minFlux) x = 0.0;
                    if (x > 0.0) x = x + addFlux;
                }
                d[c] = sat_u8_from_f64(x);
            }
        } else {
            return -3;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * A.2  equalizeHist
 * ------------------------------------------------------------------------------------------ */
int lfo_equalize_lut(const int32_t *hist, int total, uint8_t *lut, int *first_bin,
                     int *constant) {
    int i = 0;
    while (i < 256 && !hist[i]) ++i;
    if (i == 256) return -1;
    *first_bin = i;
    if (hist[i] == total) { /* single-valued image: dst.setTo(i) */
        for (int b = 0; b < 256; b++) lut[b] = (uint8_t)i;
        *constant = 1;
        return 0;
    }
    *constant = 0;
    float scale = (256 - 1.f) / (float)(total - hist[i]);
    int sum = 0;
    for (int b = 0; b <= i; b++) lut[b] = 0; /* bins below i are never read */
    for (++i; i < 256; ++i) {
        sum += hist[i];
        lut[i] = sat_u8_from_f32((float)sum * scale);
    }
    return 0;
}

int lfo_equalize_hist(const uint8_t *src, int h, int w, uint8_t *dst) {
    int32_t hist[256];
    uint8_t lut[256];
    size_t n = (size_t)h * w;
    int first, constant;
    memset(hist, 0, sizeof hist);
    for (size_t k = 0; k < n; k++) hist[src[k]]++;
    if (lfo_equalize_lut(hist, (int)n, lut, &first, &constant)) return -1;
    for (size_t k = 0; k < n; k++) dst[k] = lut[src[k]];
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * A.3  dilate / erode, anchor = (kw/2, kh/2), one iteration, border samples ignored.
 * ------------------------------------------------------------------------------------------ */
int lfo_morph(const uint8_t *src, int h, int w, const uint8_t *kernel, int kh, int kw, int op,
              uint8_t *dst) {
    if (kh <= 0 || kw <= 0) return -1;
    int ay = kh / 2, ax = kw / 2;
    int any = 0;
    for (int k = 0; k < kh * kw; k++) any |= kernel[k] != 0;
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            int acc = (op == LFO_DILATE) ? 0 : 255;
            for (int dy = 0; dy < kh; dy++) {
                int yy = y + dy - ay;
                if (yy < 0 || yy >= h) continue;
                for (int dx = 0; dx < kw; dx++) {
                    if (!kernel[dy * kw + dx]) continue;
                    int xx = x + dx - ax;
                    if (xx < 0 || xx >= w) continue;
                    int v = src[(size_t)yy * w + xx];
                    if (op == LFO_DILATE) { if (v > acc) acc = v; }
                    else { if (v < acc) acc = v; }
                }
            }
            (void)any;
            dst[(size_t)y * w + x] = (uint8_t)acc;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * A.4  Canny (aperture 3, L1 gradient, no blur)
 * ------------------------------------------------------------------------------------------ */
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

int lfo_sobel_mag(const uint8_t *src, int h, int w, int16_t *dx, int16_t *dy, int32_t *mag) {
    for (int y = 0; y < h; y++) {
        int ym = clampi(y - 1, 0, h - 1), yp = clampi(y + 1, 0, h - 1); /* BORDER_REPLICATE */
        const uint8_t *r0 = src + (size_t)ym * w, *r1 = src + (size_t)y * w,
                      *r2 = src + (size_t)yp * w;
        for (int x = 0; x < w; x++) {
            int xm = clampi(x - 1, 0, w - 1), xp = clampi(x + 1, 0, w - 1);
            int gx = (r0[xp] - r0[xm]) + 2 * (r1[xp] - r1[xm]) + (r2[xp] - r2[xm]);
            int gy = (r2[xm] - r0[xm]) + 2 * (r2[x] - r0[x]) + (r2[xp] - r0[xp]);
            size_t k = (size_t)y * w + x;
            dx[k] = (int16_t)gx;
            dy[k] = (int16_t)gy;
            mag[k] = abs(gx) + abs(gy);
        }
    }
    return 0;
}

int lfo_canny(const uint8_t *src, int h, int w, double low_thresh, double high_thresh,
              uint8_t *dst) {
    size_t n = (size_t)h * w;
    int16_t *dx = (int16_t *)malloc(n * sizeof(int16_t));
    int16_t *dy = (int16_t *)malloc(n * sizeof(int16_t));
    int32_t *mag = (int32_t *)malloc(n * sizeof(int32_t));
    uint8_t *map = (uint8_t *)malloc(n);       /* 1 = cannot be edge, 0 = maybe, 2 = edge */
    int32_t *stack = (int32_t *)malloc(n * sizeof(int32_t));
    if (!dx || !dy || !mag || !map || !stack) { free(dx); free(dy); free(mag); free(map); free(stack); return -1; }
    if (low_thresh > high_thresh) { double t = low_thresh; low_thresh = high_thresh; high_thresh = t; }
    int low = (int)floor(low_thresh), high = (int)floor(high_thresh);
    const int TG22 = 13573; /* tan(22.5 deg) * 2^15 */
    lfo_sobel_mag(src, h, w, dx, dy, mag);
    size_t sp = 0;
#define MAG(yy, xx) (((yy) < 0 || (yy) >= h || (xx) < 0 || (xx) >= w) ? 0 : mag[(size_t)(yy) * w + (xx)])
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            size_t k = (size_t)y * w + x;
            int m = mag[k];
            int keep = 0;
            if (m > low) {
                int xs = dx[k], ys = dy[k];
                int ax = abs(xs), ay = abs(ys) << 15;
                int tg22x = ax * TG22;
                if (ay < tg22x) {
                    keep = (m > MAG(y, x - 1)) && (m >= MAG(y, x + 1));
                } else {
                    int tg67x = tg22x + (ax << 16);
                    if (ay > tg67x) {
                        keep = (m > MAG(y - 1, x)) && (m >= MAG(y + 1, x));
                    } else {
                        int s = ((xs ^ ys) < 0) ? -1 : 1;
                        keep = (m > MAG(y - 1, x - s)) && (m > MAG(y + 1, x + s));
                    }
                }
            }
            if (keep) {
                if (m > high) { map[k] = 2; stack[sp++] = (int32_t)k; }
                else map[k] = 0;
            } else {
                map[k] = 1;
            }
        }
    }
#undef MAG
    while (sp) { /* hysteresis: 8-connected growth through "maybe" pixels */
        int32_t k = stack[--sp];
        int y = k / w, x = k % w;
        for (int ddy = -1; ddy <= 1; ddy++)
            for (int ddx = -1; ddx <= 1; ddx++) {
                int yy = y + ddy, xx = x + ddx;
                if ((ddy | ddx) == 0 || yy < 0 || yy >= h || xx < 0 || xx >= w) continue;
                size_t kk = (size_t)yy * w + xx;
                if (map[kk] == 0) { map[kk] = 2; stack[sp++] = (int32_t)kk; }
            }
    }
    for (size_t k = 0; k < n; k++) dst[k] = (map[k] == 2) ? 255 : 0;
    free(dx); free(dy); free(mag); free(map); free(stack);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * A.5  findContours: Suzuki-Abe border following as in OpenCV's contours.cpp
 *      (cvFindNextContour + icvFetchContour), on a copy padded by a 1-px zero frame (>= 3.2).
 * ------------------------------------------------------------------------------------------ */
typedef struct { int32_t *v; size_t n, cap; } ivec;
static int ivec_push(ivec *a, int32_t x) {
    if (a->n == a->cap) {
        size_t nc = a->cap ? a->cap * 2 : 1024;
        int32_t *nv = (int32_t *)realloc(a->v, nc * sizeof(int32_t));
        if (!nv) return -1;
        a->v = nv; a->cap = nc;
    }
    a->v[a->n++] = x;
    return 0;
}

int lfo_find_contours(const uint8_t *img, int h, int w, int mode, int32_t **points,
                      int32_t **offsets, int32_t **is_hole_out, int32_t *n_contours) {
    int ph = h + 2, pw = w + 2;
    int8_t *im = (int8_t *)calloc((size_t)ph * pw, 1);
    ivec pts = {0, 0, 0}, offs = {0, 0, 0}, holes = {0, 0, 0};
    if (!im) return -1;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) im[(size_t)(y + 1) * pw + x + 1] = img[(size_t)y * w + x] ? 1 : 0;
    /* direction codes: 0=E 1=NE 2=N 3=NW 4=W 5=SW 6=S 7=SE (y grows downwards) */
    const int cdx[8] = {1, 1, 0, -1, -1, -1, 0, 1};
    const int cdy[8] = {0, -1, -1, -1, 0, 1, 1, 1};
    int deltas[16];
    for (int k = 0; k < 16; k++) deltas[k] = cdy[k & 7] * pw + cdx[k & 7];
    const int nbd = 2;
    ivec_push(&offs, 0);
    for (int y = 1; y <= h; y++) {
        int8_t *row = im + (size_t)y * pw;
        int prev = 0;
        int lnbd_x = 0;
        for (int x = 1; x <= w; x++) {
            int p = row[x];
            if (p != prev) {
                int is_hole = 0;
                int trace = 1;
                if (!(prev == 0 && p == 1)) {
                    if (p != 0 || prev < 1) trace = 0;
                    else is_hole = 1;
                }
                if (trace && mode == LFO_RETR_EXTERNAL && (is_hole || row[lnbd_x] > 0)) trace = 0;
                if (!trace && is_hole && (prev & -2)) lnbd_x = x - 1;
                if (trace) {
                    lnbd_x = x - is_hole;
                    /* icvFetchContour from (x - is_hole, y) */
                    int8_t *i0 = row + x - is_hole, *i1, *i3, *i4 = 0;
                    int px = x - is_hole, py = y;
                    int s_end, s;
                    s_end = s = is_hole ? 0 : 4;
                    do {
                        s = (s - 1) & 7;
                        i1 = i0 + deltas[s];
                        if (*i1 != 0) break;
                    } while (s != s_end);
                    if (s == s_end) { /* isolated pixel */
                        *i0 = (int8_t)(nbd | -128);
                        ivec_push(&pts, px - 1); ivec_push(&pts, py - 1);
                    } else {
                        i3 = i0;
                        for (;;) {
                            s_end = s;
                            for (;;) {
                                i4 = i3 + deltas[++s];
                                if (*i4 != 0) break;
                            }
                            s &= 7;
                            if ((unsigned)(s - 1) < (unsigned)s_end) *i3 = (int8_t)(nbd | -128);
                            else if (*i3 == 1) *i3 = (int8_t)nbd;
                            ivec_push(&pts, px - 1); ivec_push(&pts, py - 1);
                            px += cdx[s]; py += cdy[s];
                            if (i4 == i0 && i3 == i1) break;
                            i3 = i4;
                            s = (s + 4) & 7;
                        }
                    }
                    ivec_push(&offs, (int32_t)(pts.n / 2));
                    ivec_push(&holes, is_hole);
                    p = row[x]; /* the mark just written */
                }
                prev = p;
                if (prev & -2) lnbd_x = x;
            }
        }
    }
    free(im);
    *n_contours = (int32_t)holes.n;
    if (!pts.v) pts.v = (int32_t *)malloc(8);
    if (!holes.v) holes.v = (int32_t *)malloc(8);
    *points = pts.v;
    *offsets = offs.v;
    if (is_hole_out) *is_hole_out = holes.v; else free(holes.v);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * A.6  convexHull(clockwise=true) -> rotatingCalipers -> RotatedRect -> points -> fillPoly
 * ------------------------------------------------------------------------------------------ */
static int cmp_pt(const void *a, const void *b) {
    const int32_t *p = (const int32_t *)a, *q = (const int32_t *)b;
    if (p[0] != q[0]) return p[0] < q[0] ? -1 : 1;
    if (p[1] != q[1]) return p[1] < q[1] ? -1 : 1;
    return 0;
}
static inline int64_t cross3(const int32_t *o, const int32_t *a, const int32_t *b) {
    return (int64_t)(a[0] - o[0]) * (b[1] - o[1]) - (int64_t)(a[1] - o[1]) * (b[0] - o[0]);
}

/* Strictly convex hull.  Vertex order: start at (min x, then min y), walk the large-y chain
 * to (max x, then max y), return along the small-y chain -- the order convexHull(clockwise=true)
 * yields (clockwise in a y-up frame). */
int lfo_convex_hull(const int32_t *pts_in, int n, int32_t *hull, int *n_hull) {
    if (n <= 0) { *n_hull = 0; return 0; }
    int32_t *p = (int32_t *)malloc((size_t)n * 2 * sizeof(int32_t));
    int32_t *st = (int32_t *)malloc((size_t)(n + 2) * 2 * sizeof(int32_t));
    if (!p || !st) { free(p); free(st); return -1; }
    memcpy(p, pts_in, (size_t)n * 2 * sizeof(int32_t));
    qsort(p, (size_t)n, 2 * sizeof(int32_t), cmp_pt);
    int m = 0;
    for (int i = 0; i < n; i++) /* unique */
        if (m == 0 || p[2 * i] != p[2 * m - 2] || p[2 * i + 1] != p[2 * m - 1]) {
            p[2 * m] = p[2 * i]; p[2 * m + 1] = p[2 * i + 1]; m++;
        }
    if (m == 1) { hull[0] = p[0]; hull[1] = p[1]; *n_hull = 1; free(p); free(st); return 0; }
    int k = 0, out = 0;
    for (int i = 0; i < m; i++) { /* large-y chain, left to right */
        while (k >= 2 && cross3(st + 2 * (k - 2), st + 2 * (k - 1), p + 2 * i) >= 0) k--;
        st[2 * k] = p[2 * i]; st[2 * k + 1] = p[2 * i + 1]; k++;
    }
    for (int i = 0; i < k; i++) { hull[2 * out] = st[2 * i]; hull[2 * out + 1] = st[2 * i + 1]; out++; }
    k = 0;
    for (int i = m - 1; i >= 0; i--) { /* small-y chain, right to left */
        while (k >= 2 && cross3(st + 2 * (k - 2), st + 2 * (k - 1), p + 2 * i) >= 0) k--;
        st[2 * k] = p[2 * i]; st[2 * k + 1] = p[2 * i + 1]; k++;
    }
    for (int i = 1; i < k - 1; i++) { hull[2 * out] = st[2 * i]; hull[2 * out + 1] = st[2 * i + 1]; out++; }
    *n_hull = out;
    free(p); free(st);
    return 0;
}

/* rotcalipers.cpp, CALIPERS_MINAREARECT; points as float32, n > 2; out = 3 Point2f */
static void rotating_calipers(const float *pts, int n, float *out) {
    float minarea = FLT_MAX;
    int buf_i0 = 0, buf_i5 = 0;
    float buf1 = 0, buf2 = 0, buf3 = 0, buf4 = 0;
    float *inv_len = (float *)malloc((size_t)n * 3 * sizeof(float));
    float *vect = inv_len + n; /* n (x,y) pairs */
    int left = 0, bottom = 0, right = 0, top = 0;
    int seq[4];
    float orientation = 0, base_a, base_b = 0;
    float left_x, right_x, top_y, bottom_y;
    float p0x = pts[0], p0y = pts[1];
    left_x = right_x = p0x;
    top_y = bottom_y = p0y;
    for (int i = 0; i < n; i++) {
        double dx, dy;
        if (p0x < left_x) { left_x = p0x; left = i; }
        if (p0x > right_x) { right_x = p0x; right = i; }
        if (p0y > top_y) { top_y = p0y; top = i; }
        if (p0y < bottom_y) { bottom_y = p0y; bottom = i; }
        int j = (i + 1 < n) ? i + 1 : 0;
        float p1x = pts[2 * j], p1y = pts[2 * j + 1];
        dx = p1x - p0x; /* float subtraction, widened */
        dy = p1y - p0y;
        vect[2 * i] = (float)dx;
        vect[2 * i + 1] = (float)dy;
        inv_len[i] = (float)(1. / sqrt(dx * dx + dy * dy));
        p0x = p1x; p0y = p1y;
    }
    {
        double ax = vect[2 * (n - 1)], ay = vect[2 * (n - 1) + 1];
        for (int i = 0; i < n; i++) {
            double bx = vect[2 * i], by = vect[2 * i + 1];
            double convexity = ax * by - ay * bx;
            if (convexity != 0) { orientation = (convexity > 0) ? 1.f : (-1.f); break; }
            ax = bx; ay = by;
        }
    }
    base_a = orientation;
    seq[0] = bottom; seq[1] = right; seq[2] = top; seq[3] = left;
    for (int k = 0; k < n; k++) {
        float dp[4];
        dp[0] = +base_a * vect[2 * seq[0]] + base_b * vect[2 * seq[0] + 1];
        dp[1] = -base_b * vect[2 * seq[1]] + base_a * vect[2 * seq[1] + 1];
        dp[2] = -base_a * vect[2 * seq[2]] - base_b * vect[2 * seq[2] + 1];
        dp[3] = +base_b * vect[2 * seq[3]] - base_a * vect[2 * seq[3] + 1];
        float maxcos = dp[0] * inv_len[seq[0]];
        int main_element = 0;
        for (int i = 1; i < 4; ++i) {
            float cosalpha = dp[i] * inv_len[seq[i]];
            if (cosalpha > maxcos) { main_element = i; maxcos = cosalpha; }
        }
        {
            int pindex = seq[main_element];
            float lead_x = vect[2 * pindex] * inv_len[pindex];
            float lead_y = vect[2 * pindex + 1] * inv_len[pindex];
            switch (main_element) {
            case 0: base_a = lead_x; base_b = lead_y; break;
            case 1: base_a = lead_y; base_b = -lead_x; break;
            case 2: base_a = -lead_x; base_b = -lead_y; break;
            default: base_a = -lead_y; base_b = lead_x; break;
            }
        }
        seq[main_element] += 1;
        seq[main_element] = (seq[main_element] == n) ? 0 : seq[main_element];
        {
            float dx = pts[2 * seq[1]] - pts[2 * seq[3]];
            float dy = pts[2 * seq[1] + 1] - pts[2 * seq[3] + 1];
            float width = dx * base_a + dy * base_b;
            dx = pts[2 * seq[2]] - pts[2 * seq[0]];
            dy = pts[2 * seq[2] + 1] - pts[2 * seq[0] + 1];
            float height = -dx * base_b + dy * base_a;
            float area = width * height;
            if (area <= minarea) {
                minarea = area;
                buf_i0 = seq[3]; buf1 = base_a; buf2 = width; buf3 = base_b; buf4 = height;
                buf_i5 = seq[0];
            }
        }
    }
    {
        float A1 = buf1, B1 = buf3, A2 = -buf3, B2 = buf1;
        float C1 = A1 * pts[2 * buf_i0] + pts[2 * buf_i0 + 1] * B1;
        float C2 = A2 * pts[2 * buf_i5] + pts[2 * buf_i5 + 1] * B2;
        float idet = 1.f / (A1 * B2 - A2 * B1);
        float px = (C1 * B2 - C2 * B1) * idet;
        float py = (A1 * C2 - A2 * C1) * idet;
        out[0] = px; out[1] = py;
        out[2] = A1 * buf2; out[3] = B1 * buf2;
        out[4] = A2 * buf4; out[5] = B2 * buf4;
    }
    free(inv_len);
}

/* rect = center.x, center.y, size.width, size.height, angle (degrees) */
static void min_area_rect_hull(const int32_t *hull, int n, float rect[5]) {
    float cx = 0, cy = 0, sw = 0, sh = 0, angle = 0;
    if (n > 2) {
        float *hp = (float *)malloc((size_t)n * 2 * sizeof(float));
        float out[6];
        for (int i = 0; i < 2 * n; i++) hp[i] = (float)hull[i];
        rotating_calipers(hp, n, out);
        cx = out[0] + (out[2] + out[4]) * 0.5f;
        cy = out[1] + (out[3] + out[5]) * 0.5f;
        sw = (float)sqrt((double)out[2] * out[2] + (double)out[3] * out[3]);
        sh = (float)sqrt((double)out[4] * out[4] + (double)out[5] * out[5]);
        angle = (float)atan2((double)out[3], (double)out[2]);
        free(hp);
    } else if (n == 2) {
        float x0 = (float)hull[0], y0 = (float)hull[1], x1 = (float)hull[2], y1 = (float)hull[3];
        cx = (x0 + x1) * 0.5f;
        cy = (y0 + y1) * 0.5f;
        double dx = x1 - x0, dy = y1 - y0;
        sw = (float)sqrt(dx * dx + dy * dy);
        sh = 0;
        angle = (float)atan2(dy, dx);
    } else if (n == 1) {
        cx = (float)hull[0]; cy = (float)hull[1];
    }
    angle = (float)((double)(angle * 180.0f) / LFO_PI);
    rect[0] = cx; rect[1] = cy; rect[2] = sw; rect[3] = sh; rect[4] = angle;
}

int lfo_min_area_rect(const int32_t *pts, int n, float rect[5]) {
    if (n <= 0) return -1;
    int32_t *hull = (int32_t *)malloc((size_t)n * 2 * sizeof(int32_t));
    int nh = 0;
    if (!hull) return -1;
    if (lfo_convex_hull(pts, n, hull, &nh)) { free(hull); return -1; }
    min_area_rect_hull(hull, nh, rect);
    free(hull);
    return 0;
}

void lfo_box_points(const float rect[5], float box[8]) { /* RotatedRect::points */
    double ang = rect[4] * LFO_PI / 180.;
    float b = (float)cos(ang) * 0.5f;
    float a = (float)sin(ang) * 0.5f;
    float cx = rect[0], cy = rect[1], sw = rect[2], sh = rect[3];
    box[0] = cx - a * sh - b * sw;
    box[1] = cy + b * sh - a * sw;
    box[2] = cx + a * sh - b * sw;
    box[3] = cy - b * sh - a * sw;
    box[4] = 2 * cx - box[0];
    box[5] = 2 * cy - box[1];
    box[6] = 2 * cx - box[2];
    box[7] = 2 * cy - box[3];
}

/* The libm calls of lfo_min_area_rect / lfo_box_points on caller-supplied operands, vectorised: what the GPU tests compare
 * the device's libm with (minAreaRect's angle in degrees; cos / sin of it times 0.5) */
void lfo_debug_trig(int n, const double *y, const double *x, float *angle_deg, float *cos_half, float *sin_half) {
    for (int i = 0; i < n; i++) {
        float angle = (float)atan2(y[i], x[i]);
        angle = (float)((double)(angle * 180.0f) / LFO_PI);
        double ang = angle * LFO_PI / 180.;
        angle_deg[i] = angle;
        cos_half[i] = (float)cos(ang) * 0.5f;
        sin_half[i] = (float)sin(ang) * 0.5f;
    }
}

/* drawing.cpp: clipLine on 64-bit points */
static int clip_line(int64_t width, int64_t height, int64_t *x1, int64_t *y1, int64_t *x2,
                     int64_t *y2) {
    int c1, c2;
    int64_t right = width - 1, bottom = height - 1;
    if (width <= 0 || height <= 0) return 0;
    c1 = (*x1 < 0) + (*x1 > right) * 2 + (*y1 < 0) * 4 + (*y1 > bottom) * 8;
    c2 = (*x2 < 0) + (*x2 > right) * 2 + (*y2 < 0) * 4 + (*y2 > bottom) * 8;
    if ((c1 & c2) == 0 && (c1 | c2) != 0) {
        int64_t a;
        if (c1 & 12) {
            a = c1 < 8 ? 0 : bottom;
            *x1 += (int64_t)((double)(a - *y1) * (*x2 - *x1) / (*y2 - *y1));
            *y1 = a;
            c1 = (*x1 < 0) + (*x1 > right) * 2;
        }
        if (c2 & 12) {
            a = c2 < 8 ? 0 : bottom;
            *x2 += (int64_t)((double)(a - *y2) * (*x2 - *x1) / (*y2 - *y1));
            *y2 = a;
            c2 = (*x2 < 0) + (*x2 > right) * 2;
        }
        if ((c1 & c2) == 0 && (c1 | c2) != 0) {
            if (c1) {
                a = c1 == 1 ? 0 : right;
                *y1 += (int64_t)((double)(a - *x1) * (*y2 - *y1) / (*x2 - *x1));
                *x1 = a;
                c1 = 0;
            }
            if (c2) {
                a = c2 == 1 ? 0 : right;
                *y2 += (int64_t)((double)(a - *x2) * (*y2 - *y1) / (*x2 - *x1));
                *x2 = a;
                c2 = 0;
            }
        }
    }
    return (c1 | c2) == 0;
}

/* drawing.cpp: Line() = LineIterator(connectivity 8, left_to_right) */
static void draw_line8(uint8_t *img, int h, int w, int64_t x1, int64_t y1, int64_t x2, int64_t y2,
                       uint8_t color) {
    if ((uint64_t)x1 >= (uint64_t)w || (uint64_t)x2 >= (uint64_t)w || (uint64_t)y1 >= (uint64_t)h ||
        (uint64_t)y2 >= (uint64_t)h) {
        if (!clip_line(w, h, &x1, &y1, &x2, &y2)) return;
    }
    int dx = (int)(x2 - x1), dy = (int)(y2 - y1);
    int px = (int)x1, py = (int)y1;
    if (dx < 0) { dx = -dx; dy = -dy; px = (int)x2; py = (int)y2; } /* left_to_right */
    int sy = dy < 0 ? -1 : 1;
    if (dy < 0) dy = -dy;
    int minor_x = 0, minor_y = sy, major_x = 1, major_y = 0; /* plusStep / minusStep */
    if (dy > dx) {
        int t = dx; dx = dy; dy = t;
        minor_x = 1; minor_y = 0; major_x = 0; major_y = sy;
    }
    int err = dx - (dy + dy), plusDelta = dx + dx, minusDelta = -(dy + dy), count = dx + 1;
    for (int i = 0; i < count; i++) {
        img[(size_t)py * w + px] = color;
        int mask = err < 0;
        err += minusDelta + (mask ? plusDelta : 0);
        px += major_x + (mask ? minor_x : 0);
        py += major_y + (mask ? minor_y : 0);
    }
}

/* drawing.cpp: fillPoly = CollectPolyEdges (draws the outline) + FillEdgeCollection
 * (even-odd scan conversion in 16.16 fixed point).  One polygon, shift 0, line_type 8. */
typedef struct { int y0, y1; int64_t x, dx; } poly_edge;

int lfo_fill_poly(uint8_t *img, int h, int w, const int32_t *v, int npts, uint8_t color) {
    const int XY_SHIFT = 16;
    const int64_t XY_ONE = 1 << 16;
    if (npts <= 0) return 0;
    poly_edge *edges = (poly_edge *)malloc((size_t)npts * sizeof(poly_edge));
    int total = 0;
    if (!edges) return -1;
    int64_t p0x = (int64_t)v[2 * (npts - 1)] * XY_ONE, p0y = v[2 * (npts - 1) + 1]; /* (a multiplication: << of a negative value is undefined in C99) */
    for (int i = 0; i < npts; i++) {
        int64_t p1x = (int64_t)v[2 * i] * XY_ONE, p1y = v[2 * i + 1];
        int64_t t0x = (p0x + (XY_ONE >> 1)) >> XY_SHIFT, t1x = (p1x + (XY_ONE >> 1)) >> XY_SHIFT;
        draw_line8(img, h, w, t0x, p0y, t1x, p1y, color);
        if (p0y != p1y) {
            poly_edge e;
            if (p0y < p1y) { e.y0 = (int)p0y; e.y1 = (int)p1y; e.x = p0x; }
            else { e.y0 = (int)p1y; e.y1 = (int)p0y; e.x = p1x; }
            e.dx = (p1x - p0x) / (p1y - p0y);
            edges[total++] = e;
        }
        p0x = p1x; p0y = p1y;
    }
    if (total >= 2) {
        int y_max = INT_MIN, y_min = INT_MAX;
        int64_t x_max = -1, x_min = 0x7FFFFFFFFFFFFFFFLL;
        for (int i = 0; i < total; i++) {
            poly_edge *e = &edges[i];
            int64_t xe = e->x + (int64_t)(e->y1 - e->y0) * e->dx;
            if (e->y0 < y_min) y_min = e->y0;
            if (e->y1 > y_max) y_max = e->y1;
            if (e->x < x_min) x_min = e->x;
            if (e->x > x_max) x_max = e->x;
            if (xe < x_min) x_min = xe;
            if (xe > x_max) x_max = xe;
        }
        if (!(y_max < 0 || y_min >= h || x_max < 0 || x_min >= ((int64_t)w << XY_SHIFT))) {
            if (y_max > h) y_max = h;
            for (int y = y_min; y < y_max; y++) {
                int64_t xs[64];
                int na = 0;
                for (int i = 0; i < total && na < 64; i++)
                    if (edges[i].y0 <= y && y < edges[i].y1)
                        xs[na++] = edges[i].x + (int64_t)(y - edges[i].y0) * edges[i].dx;
                for (int a = 1; a < na; a++) { /* insertion sort by x */
                    int64_t t = xs[a]; int b = a - 1;
                    while (b >= 0 && xs[b] > t) { xs[b + 1] = xs[b]; b--; }
                    xs[b + 1] = t;
                }
                if (y < 0) continue;
                for (int a = 0; a + 1 < na; a += 2) {
                    int xa = (int)((xs[a] + XY_ONE - 1) >> XY_SHIFT);
                    int xb = (int)(xs[a + 1] >> XY_SHIFT);
                    if (xa < w && xb >= 0) {
                        if (xa < 0) xa = 0;
                        if (xb >= w) xb = w - 1;
                        for (int x = xa; x <= xb; x++) img[(size_t)y * w + x] = color;
                    }
                }
            }
        }
    }
    free(edges);
    return 0;
}

/* Optional Gaussian smoothing of Canny's input (off by default; no reference call site: see lfd_oracle.h) */
int lfo_gaussian_kernel(int n, double sigma, float *cf) {
    static const float small_tab[4][7] = {{1.f}, {0.25f, 0.5f, 0.25f}, {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f},
                                          {0.03125f, 0.109375f, 0.21875f, 0.28125f, 0.21875f, 0.109375f, 0.03125f}};
    if (n <= 0 || n > 31 || (n & 1) == 0) return -1;
    const float *fixed = (n <= 7 && sigma <= 0) ? small_tab[n >> 1] : 0;
    double sigmaX = sigma > 0 ? sigma : ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
    double scale2X = -0.5 / (sigmaX * sigmaX), sum = 0;
    for (int i = 0; i < n; i++) {
        double x = i - (n - 1) * 0.5;
        double t = fixed ? (double)fixed[i] : exp(scale2X * x * x);
        cf[i] = (float)t;
        sum += cf[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < n; i++) cf[i] = (float)(cf[i] * sum);
    return 0;
}

static int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}

int lfo_gaussian_blur(const uint8_t *src, int h, int w, int ksize, double sigma, uint8_t *dst) {
    float cf[32];
    if (lfo_gaussian_kernel(ksize, sigma, cf)) return -1;
    int r = ksize / 2;
    float *tmp = (float *)malloc((size_t)h * w * sizeof(float));
    if (!tmp) return -1;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            float acc = 0.f;
            for (int k = 0; k < ksize; k++) acc = acc + (float)src[(size_t)y * w + reflect101(x + k - r, w)] * cf[k];
            tmp[(size_t)y * w + x] = acc;
        }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            float acc = 0.f;
            for (int k = 0; k < ksize; k++) acc = acc + tmp[(size_t)reflect101(y + k - r, h) * w + x] * cf[k];
            dst[(size_t)y * w + x] = sat_u8_from_f32(acc);
        }
    free(tmp);
    return 0;
}

int lfo_fit_min_area_rect(const uint8_t *img, int h, int w, int contoursMode, int contoursMethod,
                          double minAreaRectMinLen, double lwTresh, uint8_t *box_img,
                          int32_t *detection, int32_t *n_boxes) {
    return lfo_fit_min_area_rect_g(img, h, w, contoursMode, contoursMethod, minAreaRectMinLen, lwTresh, 0, 0., box_img, detection,
                                   n_boxes);
}

/* processfield.py:201-263 (gaussKernel > 0: Canny sees the smoothed image; the reference never smooths) */
int lfo_fit_min_area_rect_g(const uint8_t *img, int h, int w, int contoursMode, int contoursMethod,
                            double minAreaRectMinLen, double lwTresh, int gaussKernel, double gaussSigma,
                            uint8_t *box_img, int32_t *detection, int32_t *n_boxes) {
    size_t n = (size_t)h * w;
    uint8_t *canny = (uint8_t *)malloc(n);
    uint8_t *smooth = 0;
    if (gaussKernel > 0) {
        smooth = (uint8_t *)malloc(n);
        if (!smooth || !canny || lfo_gaussian_blur(img, h, w, gaussKernel, gaussSigma, smooth)) { free(smooth); free(canny); return -1; }
        img = smooth;
    }
    int32_t *pts = 0, *offs = 0, nc = 0;
    int det = 0, nb = 0;
    if (!canny) return -1;
    if (contoursMethod != LFO_CHAIN_APPROX_NONE && contoursMethod != LFO_CHAIN_APPROX_SIMPLE) {
        free(canny); free(smooth);
        return -4; /* TC89 approximations change the point set: not restated */
    }
    memset(box_img, 0, n);
    if (lfo_canny(img, h, w, 0, 255, canny)) { free(canny); free(smooth); return -1; }
    free(smooth);
    if (lfo_find_contours(canny, h, w, contoursMode, &pts, &offs, 0, &nc)) { free(canny); return -1; }
    for (int c = 0; c < nc; c++) {
        float rect[5];
        int cnt = offs[c + 1] - offs[c];
        lfo_min_area_rect(pts + 2 * (size_t)offs[c], cnt, rect);
        double length, width;
        if (rect[2] > rect[3]) { length = rect[2]; width = rect[3]; }
        else { width = rect[2]; length = rect[3]; }
        if (length > minAreaRectMinLen && width > minAreaRectMinLen) {
            if (length / width > lwTresh) {
                float box[8];
                int32_t ibox[8];
                det = 1; nb++;
                lfo_box_points(rect, box);
                for (int k = 0; k < 8; k++) ibox[k] = (int32_t)box[k]; /* np.int32: truncate */
                lfo_fill_poly(box_img, h, w, ibox, 4, 255);
            }
        }
    }
    free(pts); free(offs); free(canny);
    *detection = det;
    if (n_boxes) *n_boxes = nb;
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * A.7  HoughLines, classic (hough.cpp HoughLinesStandard)
 * ------------------------------------------------------------------------------------------ */
static int cv_round_d(double v) { return (int)lrint(v); }

void lfo_hough_dims(int h, int w, double rho_d, double theta_d, int *numangle, int *numrho) {
    float rho = (float)rho_d, theta = (float)theta_d;
    *numangle = cv_round_d((LFO_PI - 0.0) / theta);
    *numrho = cv_round_d(((w + h) * 2 + 1) / rho);
}

int lfo_hough_accum(const uint8_t *img, int h, int w, double rho_d, double theta_d, int32_t *accum,
                    int *numangle_out, int *numrho_out) {
    float rho = (float)rho_d, theta = (float)theta_d;
    float irho = 1 / rho;
    int numangle, numrho;
    lfo_hough_dims(h, w, rho_d, theta_d, &numangle, &numrho);
    float *tabSin = (float *)malloc((size_t)numangle * 2 * sizeof(float));
    float *tabCos = tabSin + numangle;
    if (!tabSin) return -1;
    memset(accum, 0, (size_t)(numangle + 2) * (numrho + 2) * sizeof(int32_t));
    float ang = 0.f;
    for (int n = 0; n < numangle; ang += theta, n++) {
        tabSin[n] = (float)(sin((double)ang) * irho);
        tabCos[n] = (float)(cos((double)ang) * irho);
    }
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++)
            if (img[(size_t)i * w + j] != 0)
                for (int n = 0; n < numangle; n++) {
                    float fj = (float)j * tabCos[n], fi = (float)i * tabSin[n];
                    int r = (int)lrintf(fj + fi);
                    r += (numrho - 1) / 2;
                    accum[(n + 1) * (numrho + 2) + r + 1]++;
                }
    free(tabSin);
    *numangle_out = numangle;
    *numrho_out = numrho;
    return 0;
}

static __thread const int32_t *g_sort_accum; /* thread-local: bench.py and the tests call the oracle from several threads */
static int cmp_hough(const void *a, const void *b) {
    int l1 = *(const int32_t *)a, l2 = *(const int32_t *)b;
    if (g_sort_accum[l1] != g_sort_accum[l2]) return g_sort_accum[l1] > g_sort_accum[l2] ? -1 : 1;
    return l1 < l2 ? -1 : (l1 > l2 ? 1 : 0);
}

int lfo_hough_lines(const uint8_t *img, int h, int w, double rho_d, double theta_d, int threshold,
                    int max_lines, float *lines, int32_t *n_lines) {
    float rho = (float)rho_d, theta = (float)theta_d;
    int numangle, numrho;
    lfo_hough_dims(h, w, rho_d, theta_d, &numangle, &numrho);
    if (numangle <= 0 || numrho <= 0) return -1;
    size_t na = (size_t)(numangle + 2) * (numrho + 2);
    int32_t *accum = (int32_t *)malloc(na * sizeof(int32_t));
    int32_t *sort_buf = (int32_t *)malloc(na * sizeof(int32_t));
    int total = 0;
    if (!accum || !sort_buf) { free(accum); free(sort_buf); return -1; }
    lfo_hough_accum(img, h, w, rho_d, theta_d, accum, &numangle, &numrho);
    for (int r = 0; r < numrho; r++)
        for (int n = 0; n < numangle; n++) {
            int base = (n + 1) * (numrho + 2) + r + 1;
            if (accum[base] > threshold && accum[base] > accum[base - 1] &&
                accum[base] >= accum[base + 1] && accum[base] > accum[base - numrho - 2] &&
                accum[base] >= accum[base + numrho + 2])
                sort_buf[total++] = base;
        }
    g_sort_accum = accum;
    qsort(sort_buf, (size_t)total, sizeof(int32_t), cmp_hough);
    int nout = total < max_lines ? total : max_lines;
    float scale = 1.f / (numrho + 2);
    for (int i = 0; i < nout; i++) {
        int idx = sort_buf[i];
        int n = (int)floorf((float)idx * scale) - 1; /* cvFloor(idx*scale) - 1 */
        int r = idx - (n + 1) * (numrho + 2) - 1;
        lines[2 * i] = (r - (numrho - 1) * 0.5f) * rho;
        lines[2 * i + 1] = 0.f + n * theta;
    }
    *n_lines = total;
    free(accum); free(sort_buf);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * processfield.py:36-150  check_theta.  float64 arrays of length navg, zero-filled; entries
 * are copied while BOTH sets still have line i (the IndexError `pass` skips the rest of the
 * loop body, and rho1[i] is assigned before hough2[i] is touched).
 * ------------------------------------------------------------------------------------------ */
/* numpy add.reduce on a contiguous float64 vector: first element + pairwise_sum(rest) */
static double np_pairwise(const double *a, int n) {
    if (n < 8) {
        double res = 0.;
        for (int i = 0; i < n; i++) res += a[i];
        return res;
    } else if (n <= 128) {
        double r[8], res;
        int i;
        for (int j = 0; j < 8; j++) r[j] = a[j];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        int n2 = n / 2;
        n2 -= n2 % 8;
        return np_pairwise(a, n2) + np_pairwise(a + n2, n - n2);
    }
}
static double np_mean(const double *a, int n) { return (a[0] + np_pairwise(a + 1, n - 1)) / n; }

int lfo_check_theta(const float *h1, int n1, const float *h2, int n2, int navg, double dro,
                    double thetaTresh, double lineSetTresh) {
    if (navg <= 0) return 0;
    double *ro1 = (double *)calloc((size_t)navg * 4, sizeof(double));
    double *ro2 = ro1 + navg, *t1 = ro2 + navg, *t2 = t1 + navg;
    int ret = 0;
    for (int i = 0; i < navg; i++) {
        /* order of the four assignments: ro1, ro2, theta1, theta2 (processfield.py:97-100) */
        if (i >= n1) continue;
        ro1[i] = h1[2 * i];
        if (i >= n2) continue;
        ro2[i] = h2[2 * i];
        t1[i] = h1[2 * i + 1];
        t2[i] = h2[2 * i + 1];
    }
    if (fabs(np_mean(ro1, navg) - np_mean(ro2, navg)) > dro) ret = 1;
    if (!ret) {
        double mx = t1[0], mn = t1[0];
        for (int i = 1; i < navg; i++) { if (t1[i] > mx) mx = t1[i]; if (t1[i] < mn) mn = t1[i]; }
        if (fabs(mx - mn) > thetaTresh) ret = 1;
    }
    if (!ret) {
        double mx = t2[0], mn = t2[0];
        for (int i = 1; i < navg; i++) { if (t2[i] > mx) mx = t2[i]; if (t2[i] < mn) mn = t2[i]; }
        if (fabs(mx - mn) > thetaTresh) ret = 1;
    }
    if (!ret) {
        for (int i = 0; i < navg; i++) ro1[i] = fabs(t1[i] - t2[i]);
        if (np_mean(ro1, navg) > lineSetTresh) ret = 1;
    }
    free(ro1);
    return ret;
}

/* processfield.py:266-288, float32 scalar arithmetic (NumPy >= 2 promotion rules) */
void lfo_dictify_hough(int n_x, int n_y, float rho, float theta, int32_t out[4]) {
    float c = cosf(theta), s = sinf(theta);
    float x0 = c * rho, y0 = s * rho;
    float L = (float)(n_x + n_y);
    out[0] = (int32_t)(x0 - L * s);
    out[1] = (int32_t)(y0 + L * c);
    out[2] = (int32_t)(x0 + L * s);
    out[3] = (int32_t)(y0 - L * c);
}

/* ------------------------------------------------------------------------------------------
 * removestars.py:111-130 (math.ceil of every column) and :212-231 (tests + axis-swapped fill
 * with Python slice semantics).
 * ------------------------------------------------------------------------------------------ */
static void py_slice(long start, long stop, long len, long *a, long *b) {
    if (start < 0) { start += len; if (start < 0) start = 0; } else if (start > len) start = len;
    if (stop < 0) { stop += len; if (stop < 0) stop = 0; } else if (stop > len) stop = len;
    *a = start; *b = stop;
}

int lfo_remove_stars(float *img, int h, int w, int n_obj, const float *rowc, const float *colc,
                     const float *psfmag, const float *petro90, const int32_t *nobserve,
                     const int32_t *ndetect, const lfo_rs_params *p) {
    int f = p->filter_index;
    if (f < 0 || f > 4) return -1;
    for (int i = 0; i < n_obj; i++) {
        long x = (long)ceil((double)colc[5 * i + f]); /* x = COLC, used on axis 0 (F6) */
        long y = (long)ceil((double)rowc[5 * i + f]);
        long mags[5];
        for (int k = 0; k < 5; k++) mags[k] = (long)ceil((double)psfmag[5 * i + k]);
        if (!((double)mags[f] < p->filter_cap)) continue;
        int cnt = 0;
        for (int j = 0; j < 5; j++)
            for (int k = j + 1; k < 5; k++) {
                long d = mags[j] - mags[k];
                if (d < 0) d = -d;
                if ((double)d > p->maxmagdiff) cnt++;
            }
        if (!(p->magcount >= cnt)) continue;
        long dxy = p->defaultxy;
        long pet = (long)ceil((double)petro90[5 * i + f]);
        if (pet > 0) dxy = (long)((double)pet / p->pixscale) + 10;
        if (dxy > p->maxxy) dxy = p->defaultxy;
        if (nobserve[i] != ndetect[i]) continue;
        long r0, r1, c0, c1;
        py_slice(x - dxy, x + dxy, h, &r0, &r1);
        py_slice(y - dxy, y + dxy, w, &c0, &c1);
        for (long r = r0; r < r1; r++)
            for (long c = c0; c < c1; c++) img[(size_t)r * w + c] = 0.0f;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * processfield.py:291-388 / :391-506
 * ------------------------------------------------------------------------------------------ */
static int detect_tail(const uint8_t *equ, int h, int w, const lfo_params *p, lfo_result *res,
                       uint8_t *box_img, int which) {
    int32_t det = 0, nb = 0;
    int rc = lfo_fit_min_area_rect_g(equ, h, w, p->contoursMode, p->contoursMethod, p->minAreaRectMinLen, p->lwTresh,
                                     p->gaussKernel, p->gaussSigma, box_img, &det, &nb);
    if (rc) return rc;
    res->detection = det;
    res->rejected_by_theta = 0;
    res->n_lines_equ = res->n_lines_box = 0;
    if (!det) return 0;
    int k = p->nlinesInSet > 0 ? p->nlinesInSet : 1;
    float *l1 = (float *)malloc((size_t)k * 4 * sizeof(float));
    float *l2 = l1 + 2 * k;
    int32_t n1 = 0, n2 = 0;
    double theta = LFO_PI / 180;
    lfo_hough_lines(equ, h, w, p->houghMethod, theta, 1, k, l1, &n1);
    lfo_hough_lines(box_img, h, w, p->houghMethod, theta, 1, k, l2, &n2);
    res->n_lines_equ = n1; res->n_lines_box = n2;
    if (n1 == 0 || n2 == 0) { /* HoughLines returned None -> TypeError in check_theta */
        res->status = -1;
        free(l1);
        return 0;
    }
    int m1 = n1 < k ? n1 : k, m2 = n2 < k ? n2 : k;
    if (lfo_check_theta(l1, m1, l2, m2, p->nlinesInSet, p->dro, p->thetaTresh, p->lineSetTresh)) {
        res->rejected_by_theta = 1;
    } else {
        int32_t o[4];
        res->found = which;
        res->rho = l1[0]; res->theta = l1[1];
        lfo_dictify_hough(h, w, l1[0], l1[1], o);
        res->x1 = o[0]; res->y1 = o[1]; res->x2 = o[2]; res->y2 = o[3];
    }
    free(l1);
    return 0;
}

int lfo_process_bright(const void *img, int dtype, int h, int w, int flip, const lfo_params *p,
                       lfo_result *res, uint8_t *equ_out, uint8_t *box_out) {
    size_t n = (size_t)h * w;
    uint8_t *gray = (uint8_t *)malloc(n), *equ = (uint8_t *)malloc(n), *dil = (uint8_t *)malloc(n),
            *box = (uint8_t *)malloc(n);
    int rc = -1;
    memset(res, 0, sizeof *res);
    if (!gray || !equ || !dil || !box) goto done;
    if ((rc = lfo_prep(img, dtype, h, w, flip, LFO_PREP_BRIGHT, 0, 0, gray))) goto done;
    if ((rc = lfo_equalize_hist(gray, h, w, equ))) goto done;
    if ((rc = lfo_morph(equ, h, w, p->dilateKernel, p->dilate_kh, p->dilate_kw, LFO_DILATE, dil))) goto done;
    rc = detect_tail(dil, h, w, p, res, box, 1);
    if (equ_out) memcpy(equ_out, dil, n);
    if (box_out) memcpy(box_out, box, n);
done:
    free(gray); free(equ); free(dil); free(box);
    if (rc) res->status = rc;
    return rc;
}

int lfo_process_dim(const void *img, int dtype, int h, int w, int flip, int after_bright,
                    const lfo_params *p, lfo_result *res, uint8_t *equ_out, uint8_t *box_out) {
    size_t n = (size_t)h * w;
    uint8_t *gray = (uint8_t *)malloc(n), *equ = (uint8_t *)malloc(n), *ero = (uint8_t *)malloc(n),
            *dil = (uint8_t *)malloc(n), *box = (uint8_t *)malloc(n);
    int rc = -1;
    memset(res, 0, sizeof *res);
    if (!gray || !equ || !ero || !dil || !box) goto done;
    if ((rc = lfo_prep(img, dtype, h, w, flip, after_bright ? LFO_PREP_BRIGHT_THEN_DIM : LFO_PREP_DIM,
                       p->minFlux, p->addFlux, gray))) goto done;
    if ((rc = lfo_equalize_hist(gray, h, w, equ))) goto done;
    if ((rc = lfo_morph(equ, h, w, p->erodeKernel, p->erode_kh, p->erode_kw, LFO_ERODE, ero))) goto done;
    if ((rc = lfo_morph(ero, h, w, p->dilateKernel, p->dilate_kh, p->dilate_kw, LFO_DILATE, dil))) goto done;
    rc = detect_tail(dil, h, w, p, res, box, 2);
    if (equ_out) memcpy(equ_out, dil, n);
    if (box_out) memcpy(box_out, box, n);
done:
    free(gray); free(equ); free(ero); free(dil); free(box);
    if (rc) res->status = rc;
    return rc;
}

/* detecttrails.py:119-131: remove_stars -> flip -> bright -> (dim if bright failed) */
int lfo_detect_frame(float *img, int h, int w, const lfo_params *bright, const lfo_params *dim,
                     int n_obj, const float *rowc, const float *colc, const float *psfmag,
                     const float *petro90, const int32_t *nobserve, const int32_t *ndetect,
                     const lfo_rs_params *rs, lfo_result *res) {
    int rc;
    if (n_obj > 0 && rs) {
        rc = lfo_remove_stars(img, h, w, n_obj, rowc, colc, psfmag, petro90, nobserve, ndetect, rs);
        if (rc) { memset(res, 0, sizeof *res); res->status = rc; return rc; }
    }
    rc = lfo_process_bright(img, LFO_F32, h, w, 1, bright, res, 0, 0);
    if (rc || res->status || res->found) return rc;
    return lfo_process_dim(img, LFO_F32, h, w, 1, 1, dim, res, 0, 0);
}
