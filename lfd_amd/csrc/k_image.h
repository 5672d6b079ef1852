// k_image.h -- dense image stages: remove_stars fill, prep (flip + mask + convertScaleAbs +
// histogram), equalisation LUT, rectangular / generic erode-dilate, Sobel+NMS.
// Reference call sites: removestars.py:212-231, detecttrails.py:124, processfield.py:342-354,
// :453-471, :236 (Canny's first half).  All HBM-bound; algorithmic bytes in DESIGN.md.
#pragma once
#include <type_traits>
#include "common.h"
#include "../../include/lfdmi.h"

// ------------------------------------------------------------------------------------------
// remove_stars: the catalogue tests of removestars.py:213-230 (math.ceil of every column, cap test,
// pairwise-difference count, Petrosian square size, NOBSERVE == NDETECT) per object, then the zero fill of
// the axis-swapped, Python-slice-clipped square img[x-d:x+d, y-d:y+d] (x = COLC on axis 0).
// ------------------------------------------------------------------------------------------
struct RsDev {
    int defaultxy, maxxy, magcount, filter_index;
    double pixscale, maxmagdiff, filter_cap;
};

__device__ __forceinline__ void py_slice(long start, long stop, long len, int *a, int *b) {
    if (start < 0) { start += len; if (start < 0) start = 0; } else if (start > len) start = len;
    if (stop < 0) { stop += len; if (stop < 0) stop = 0; } else if (stop > len) stop = len;
    *a = (int)start; *b = (int)stop;
}

// Two kernels.  k_rs_boxes: a LANE per catalogue object evaluates the tests (coalesced column loads, every lane busy) and leaves
// the square to blot -- rows [x, y) x columns [z, w), empty for an object that stays -- in `boxes`.  k_rs_fill: a WAVE per
// object (four per workgroup, no barrier) zero-fills its square row by row.  (One kernel doing both kept 63 lanes of every
// wave waiting behind lane 0's chain of dependent column loads and double-precision tests before the first store.)
__global__ void __launch_bounds__(256)
k_rs_boxes(int h, int w, int max_obj, const int *count, const float *rowc, const float *colc, const float *psfmag,
           const float *petro90, const int *nobserve, const int *ndetect, RsDev p, int4 *boxes) {
    const int f = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= max_obj) return;
    int r0 = 0, r1 = 0, c0 = 0, c1 = 0;
    if (i < count[f]) {
        size_t o5 = ((size_t)f * max_obj + i) * 5, o1 = (size_t)f * max_obj + i;
        int fi = p.filter_index;
        long x = (long)ceil((double)colc[o5 + fi]);
        long y = (long)ceil((double)rowc[o5 + fi]);
        long mags[5];
        for (int k = 0; k < 5; k++) mags[k] = (long)ceil((double)psfmag[o5 + k]);
        bool ok = (double)mags[fi] < p.filter_cap;
        int cnt = 0;
        for (int j = 0; j < 5; j++)
            for (int k = j + 1; k < 5; k++) {
                long d = mags[j] - mags[k];
                if (d < 0) d = -d;
                if ((double)d > p.maxmagdiff) cnt++;
            }
        ok = ok && (p.magcount >= cnt);
        long dxy = p.defaultxy;
        long pet = (long)ceil((double)petro90[o5 + fi]);
        if (pet > 0) dxy = (long)((double)pet / p.pixscale) + 10;
        if (dxy > p.maxxy) dxy = p.defaultxy;
        ok = ok && (nobserve[o1] == ndetect[o1]);
        if (ok) {
            py_slice(x - dxy, x + dxy, h, &r0, &r1);
            py_slice(y - dxy, y + dxy, w, &c0, &c1);
        }
    }
    boxes[(size_t)f * max_obj + i] = make_int4(r0, r1, c0, c1);
}

__global__ void __launch_bounds__(256)
k_rs_fill(float *frames, int h, int w, int max_obj, const int *count, const int4 *boxes, int bx0) { // bx0: first block of objects (the fill can be launched in parts)
    int f = blockIdx.y, i = (blockIdx.x + bx0) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= max_obj) return;
    const int4 bx = boxes[(size_t)f * max_obj + i]; // (one address per wave: a broadcast load)
    const int r0 = __builtin_amdgcn_readfirstlane(bx.x), r1 = __builtin_amdgcn_readfirstlane(bx.y);
    const int c0 = __builtin_amdgcn_readfirstlane(bx.z), c1 = __builtin_amdgcn_readfirstlane(bx.w);
    int nc = c1 - c0;
    if (r1 <= r0 || nc <= 0) return;
    float *img = frames + (size_t)f * h * w;
    if (nc <= 32) { // two rows per step
        for (int r = r0 + (lane >> 5); r < r1; r += 2)
            if ((lane & 31) < nc) img[(size_t)r * w + c0 + (lane & 31)] = 0.0f;
    } else {
        for (int r = r0; r < r1; r++)
            for (int c = c0 + lane; c < c1; c += 64) img[(size_t)r * w + c] = 0.0f;
    }
}

// Crowded catalogues (thousands of objects per frame).  The kernels above cost (objects) x (something per object): fine for
// the few hundred stars of a high-latitude field, but the sweep's fold looks at EVERY object in every eight-row workgroup
// (objects x 187 per SDSS frame) and the fill writes every square whether or not another one has blotted the pixels already
// (20 000 stars: ten times the frame's area).  k_rs_sort orders a frame's non-empty squares by their first row (counting
// sort, one workgroup per frame) and leaves rowstart[r] = squares starting above row r; a square is at most `hmax` rows
// high, so the squares that can touch rows [a, b) are the contiguous range rowstart[a - hmax] .. rowstart[b] of the sorted
// list.  The fold walks that range only, and k_rs_fill_bands blots a band of rows through the same LDS bit plane: every
// pixel is written once.  Used when a chunk's catalogue holds more than RS_SORT_MIN objects per frame (host: run_removestars).
#define RS_SORT_MIN 1024
#define RS_SORT_MAXH 8192
__global__ void __launch_bounds__(1024)
k_rs_sort(int h, int max_obj, const int *count, const int4 *boxes, int4 *sboxes, int *rowstart) {
    const int f = blockIdx.x, n = min(count[f], max_obj);
    __shared__ int cur[RS_SORT_MAXH + 1];
    __shared__ int wsum[16];
    const int4 *bx = boxes + (size_t)f * max_obj;
    for (int r = threadIdx.x; r <= h; r += 1024) cur[r] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 1024) {
        const int4 b = bx[i];
        if (b.y > b.x && b.w > b.z) atomicAdd(&cur[b.x], 1);
    }
    __syncthreads();
    // exclusive scan of cur[0 .. h]: 8 consecutive rows per thread (h + 1 <= 8192), then across the workgroup
    const int per = (h + 1 + 1023) / 1024, r0 = threadIdx.x * per, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int loc = 0;
    for (int k = 0; k < per; k++) if (r0 + k <= h) loc += cur[r0 + k];
    int inc = loc;
    for (int off = 1; off < 64; off <<= 1) { int t = __shfl_up(inc, off); if (lane >= off) inc += t; }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    int base = inc - loc;
    for (int k = 0; k < wv; k++) base += wsum[k];
    for (int k = 0; k < per; k++)
        if (r0 + k <= h) {
            const int c = cur[r0 + k];
            cur[r0 + k] = base;
            rowstart[(size_t)f * (h + 1) + r0 + k] = base;
            base += c;
        }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 1024) {
        const int4 b = bx[i];
        if (b.y > b.x && b.w > b.z) sboxes[(size_t)f * max_obj + atomicAdd(&cur[b.x], 1)] = b;
    }
}

// marks the squares of sorted range [lo, hi) that cross source rows [sa, sb) in the bit plane `rsm` (rsw words per row; plane row
// of source row sr: flip ? top - sr : sr - sa, with top = sb - 1)
__device__ __forceinline__ void rs_mark(uint32_t *rsm, int rsw, const int4 *sb_, int lo, int hi, int sa, int sb, bool flip, int nthreads) {
    for (int o = lo + threadIdx.x; o < hi; o += nthreads) {
        const int4 bx = sb_[o]; // rows [x, y) x columns [z, w)
        const int y0 = max(bx.x, sa), y1 = min(bx.y, sb);
        if (y0 >= y1 || bx.w <= bx.z) continue;
        const int wa = bx.z >> 5, wb = (bx.w - 1) >> 5;
        for (int sr = y0; sr < y1; sr++) {
            uint32_t *row = rsm + (flip ? sb - 1 - sr : sr - sa) * rsw;
            for (int wq_ = wa; wq_ <= wb; wq_++) {
                uint32_t m = 0xFFFFFFFFu;
                if (wq_ == wa) m &= 0xFFFFFFFFu << (bx.z & 31);
                if (wq_ == wb) m &= 0xFFFFFFFFu >> (31 - ((bx.w - 1) & 31));
                atomicOr(&row[wq_], m);
            }
        }
    }
}

#define RS_BAND_ROWS 8
__global__ void __launch_bounds__(256)
k_rs_fill_bands(float *frames, int h, int w, int max_obj, const int4 *sboxes, const int *rowstart, int hmax) { // w % 32 == 0, w <= RS_MAXW
    const int f = blockIdx.y, ra = blockIdx.x * RS_BAND_ROWS, rb = min(ra + RS_BAND_ROWS, h);
    __shared__ uint32_t rsm[RS_BAND_ROWS * (4096 / 32)];
    const int rsw = w >> 5;
    const int *rs_ = rowstart + (size_t)f * (h + 1);
    const int lo = rs_[max(0, ra - hmax)], hi = rs_[rb];
    if (lo >= hi) return; // (uniform: no square reaches this band)
    for (int k = threadIdx.x; k < RS_BAND_ROWS * rsw; k += 256) rsm[k] = 0u;
    __syncthreads();
    rs_mark(rsm, rsw, sboxes + (size_t)f * max_obj, lo, hi, ra, rb, false, 256);
    __syncthreads();
    float *img = frames + (size_t)f * h * w;
    const int groups = w >> 2; // four pixels per thread and step: a 16-byte store where all four are blotted
    for (int k = threadIdx.x; k < (rb - ra) * groups; k += 256) {
        const int r = k / groups, gq = k - r * groups;
        const unsigned nib = (rsm[r * rsw + (gq >> 3)] >> (4 * (gq & 7))) & 0xFu;
        if (!nib) continue;
        float *p = img + (size_t)(ra + r) * w + 4 * gq;
        if (nib == 0xFu) *(float4 *)p = make_float4(0.f, 0.f, 0.f, 0.f);
        else for (int j = 0; j < 4; j++) if ((nib >> j) & 1u) p[j] = 0.0f;
    }
}

// ------------------------------------------------------------------------------------------
// prep: gray = saturate_u8(round_half_even(|mask(x)|)), rows optionally flipped, plus the
// 256-bin histogram equalizeHist needs.  16 B per lane loads (4 x f32), packed 4 x u8 stores.
// mode bit 0: x<0 -> 0 (bright); bit 1: x<minFlux -> 0, x>0 -> x+addFlux (dim).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned sat_u8_f32(float x) {
    // saturate_cast<uchar>(|x|) with round-half-even: v_rndne_f32, v_cvt_u32_f32, v_min_u32.  The conversion is written as
    // the instruction itself: its hardware result for NaN (0) and for values beyond 32 bits (0xFFFFFFFF) is what is wanted
    // here, whereas a C++ float -> unsigned cast of such values is undefined.
    unsigned u;
    asm("v_cvt_u32_f32 %0, %1" : "=v"(u) : "v"(__builtin_rintf(fabsf(x))));
    return min(u, 255u);
}
__device__ __forceinline__ unsigned sat_u8_f64(double x) {
    double a = fabs(x);
    if (a != a) return 0u;
    if (a >= 255.5) return 255u;
    return (unsigned)__double2int_rn(a);
}
__device__ __forceinline__ unsigned prep_f32(float x, int mode, float mf, float af) {
    if (mode & 1) { if (x < 0.0f) x = 0.0f; }
    if (mode & 2) { if (x < mf) x = 0.0f; if (x > 0.0f) x = __fadd_rn(x, af); }
    return sat_u8_f32(x);
}
// the same with the mode known at compile time and selects instead of branches (the float kernels are bound by their
// instruction stream, not by HBM: every instruction per pixel counts)
template <int MODE>
__device__ __forceinline__ unsigned prep_f32_m(float x, float mf, float af) {
    if (MODE & 1) x = x < 0.0f ? 0.0f : x;
    if (MODE & 2) { x = x < mf ? 0.0f : x; x = x > 0.0f ? __fadd_rn(x, af) : x; }
    return sat_u8_f32(x);
}
// ... and four values at once.  (V_CVT_PK_U8_F32, which saturates and drops the byte into place in one instruction, measured
// no faster than v_cvt_u32_f32 + v_min_u32 + the shifts here and 2 % slower in the HBM-bound 4096 x 4096 sweep.)
template <int MODE>
__device__ __forceinline__ float prep_mask_m(float x, float mf, float af) {
    if (MODE & 1) x = x < 0.0f ? 0.0f : x;
    if (MODE & 2) { x = x < mf ? 0.0f : x; x = x > 0.0f ? __fadd_rn(x, af) : x; }
    return x;
}
__device__ __forceinline__ uint32_t pack_sat_u8x4(float a, float b, float c, float d) {
    return sat_u8_f32(a) | (sat_u8_f32(b) << 8) | (sat_u8_f32(c) << 16) | (sat_u8_f32(d) << 24);
}
template <int MODE>
__device__ __forceinline__ uint32_t prep_word_m(float4 v, float mf, float af) {
    return pack_sat_u8x4(prep_mask_m<MODE>(v.x, mf, af), prep_mask_m<MODE>(v.y, mf, af), prep_mask_m<MODE>(v.z, mf, af),
                         prep_mask_m<MODE>(v.w, mf, af));
}
// bit 0 of each of a word's four bytes gathered into a nibble (byte j -> bit j): one V_DOT4_U32_U8
__device__ __forceinline__ uint32_t lsb_nibble(uint32_t w01) { return __builtin_amdgcn_udot4(w01, 0x08040201u, 0u, false); }
// Histogram of four converted pixels (one packed word).  Sky frames are zeros (bright pass) or zeros and ones (dim pass:
// 78 % / 22 %): words made of those two values are counted in registers (n01 words, ones01 one-bytes among them), only
// the others touch the LDS histogram.
struct HistAcc { int n01 = 0, ones01 = 0, zeros = 0, ones = 0; };
__device__ __forceinline__ void hist_word(uint32_t word, int *shrow, HistAcc &A) {
    const bool f01 = (word & 0xFEFEFEFEu) == 0u;
    A.n01 += f01 ? 1 : 0;
    A.ones01 += f01 ? __popc(word) : 0;
    if (f01) return;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        unsigned v = (word >> (8 * q)) & 0xffu;
        if (v > 1u) atomicAdd(&shrow[v], 1);
        else if (v) A.ones++;
        else A.zeros++;
    }
}
__device__ __forceinline__ void hist_flush(HistAcc &A, int *shrow) { // one LDS add per wave and value
    int z = 4 * A.n01 - A.ones01 + A.zeros, o = A.ones01 + A.ones;
    for (int off = 32; off > 0; off >>= 1) { z += __shfl_down(z, off); o += __shfl_down(o, off); }
    if (lfd_lane() == 0 && z) atomicAdd(&shrow[0], z);
    if (lfd_lane() == 0 && o) atomicAdd(&shrow[1], o);
}
__device__ __forceinline__ bool no_zero_byte(uint32_t w) { return ((w - 0x01010101u) & ~w & 0x80808080u) == 0u; }
__device__ __forceinline__ unsigned prep_f64(double x, int mode, double mf, double af) {
    if (mode & 1) { if (x < 0.0) x = 0.0; }
    if (mode & 2) { if (x < mf) x = 0.0; if (x > 0.0) x = __dadd_rn(x, af); }
    return sat_u8_f64(x);
}


// Cell occupancy bitmap handed from the prep kernels to k_dilate_canny_w: per band of 16 image rows
// CELLBM_WORDS u64, bit c = "some byte of columns 16 c .. 16 c + 15 of this band is non-zero" (a superset is
// fine: the consumer only skips tiles whose cells are all clear and checks the loaded bytes otherwise).
#define CELLBM_ROWS 16
#define CELLBM_COLS 16
#define CELLBM_WORDS 8 // 512 cells: image widths up to 8191

// MODE 0 .. 3: float32 frames with the mode known at compile time; -1: any dtype, run-time mode.
// DELTA (lfdmi_detect_batch's bright pass): the same sweep also evaluates the DIM pass's conversion (mode 3 with mf2 / af2) of
// every pixel.  With 0 <= addFlux <= 1 and minFlux <= 0.5 that value is the bright one or one more, so one bit per pixel
// (dbits, a bit-row plane in the output orientation) and the dim image's histogram (hist2) are all the dim pass's front end
// needs besides the 8-bit bright image: it never reads the float frames again (12.2 MB -> 3.4 MB per SDSS frame).
// RS: remove_stars' squares (k_rs_boxes) are applied to the values as they are loaded -- a blotted pixel is 0.0f whatever the
// frame holds -- so the sweep does not wait for k_rs_fill's stores (which then run beside the latency-bound stages of the
// pass, see lfdmi_detect_batch).  A workgroup marks the squares that cross its rows in an LDS bit plane first (every object
// of the frame is looked at: ~400 int4 from L2, two per lane), one bit per pixel, rows of RS_MAXW / 32 words.
#define RS_MAXW 4096
#define RS_MAXROWS 16
template <int MODE, bool DELTA = false, bool MFPOS = false, bool RS = false> // MFPOS: DELTA with mf2 > 0 (one compare and one select per value)
__global__ void __launch_bounds__(256)
k_prep_hist(const void *src, int dtype, int h, int w, int flip, int mode, double minFlux,
            double addFlux, uint8_t *gray, int *hist, u64 *cellbm, int bm_bands, const int *active, u64 *fullbits,
            int prep_rows, // rows per workgroup: a divisor of CELLBM_ROWS (a workgroup's rows lie in one band)
            u64 *dbits = nullptr, int *hist2 = nullptr, float mf2 = 0.f, float af2 = 0.f, u64 *nzd = nullptr,
            int sky_fast = 0, // DELTA + MFPOS with fl(mf2 + af2) > 0.5 (host-checked): the all-sky shortcut below is exact
            const int4 *rs_boxes = nullptr, const int *rs_count = nullptr, int rs_max_obj = 0, // RS: w <= RS_MAXW, prep_rows <= RS_MAXROWS
            const int *rs_rowstart = nullptr, int rs_hmax = 0) { // (crowded catalogues: rs_boxes sorted by first row, see k_rs_sort)
    int g = blockIdx.y;
    if (active && !active[g]) return;
    __shared__ int sh[DELTA ? 8 : 4][256];
    __shared__ uint32_t rsm[RS ? RS_MAXROWS * (RS_MAXW / 32) : 1];
    const int rsw = w >> 5; // (RS: w % 32 == 0, a DELTA precondition)
    for (int k = threadIdx.x; k < (DELTA ? 2048 : 1024); k += 256) ((int *)sh)[k] = 0;
    if (RS) {
        for (int k = threadIdx.x; k < prep_rows * rsw; k += 256) rsm[k] = 0u;
        __syncthreads();
        const int ra = blockIdx.x * prep_rows, rb_ = min(ra + prep_rows, h); // this workgroup's output rows [ra, rb_)
        // ... which are the source rows [sa, sb) (k_rs_boxes' squares are in the frame's own orientation)
        const int sa = flip ? h - rb_ : ra, sb = flip ? h - ra : rb_;
        if (rs_rowstart) {
            const int *rs_ = rs_rowstart + (size_t)g * (h + 1);
            rs_mark(rsm, rsw, rs_boxes + (size_t)g * rs_max_obj, rs_[max(0, sa - rs_hmax)], rs_[sb], sa, sb, flip != 0, 256);
        } else {
        const int n_obj = min(rs_count[g], rs_max_obj);
        for (int o = threadIdx.x; o < n_obj; o += 256) {
            const int4 bx = rs_boxes[(size_t)g * rs_max_obj + o]; // rows [x, y) x columns [z, w)
            const int y0 = max(bx.x, sa), y1 = min(bx.y, sb);
            if (y0 >= y1 || bx.w <= bx.z) continue;
            const int wa = bx.z >> 5, wb = (bx.w - 1) >> 5;
            for (int sr = y0; sr < y1; sr++) {
                const int r = flip ? h - 1 - sr : sr;
                uint32_t *row = rsm + (r - ra) * rsw;
                for (int wq_ = wa; wq_ <= wb; wq_++) {
                    uint32_t m = 0xFFFFFFFFu;
                    if (wq_ == wa) m &= 0xFFFFFFFFu << (bx.z & 31);
                    if (wq_ == wb) m &= 0xFFFFFFFFu >> (31 - ((bx.w - 1) & 31));
                    atomicOr(&row[wq_], m);
                }
            }
        }
        }
    }
    __syncthreads();
    int wv = threadIdx.x >> 6;
    size_t N = (size_t)h * w;
    uint8_t *gout = gray + (size_t)g * N;
    int zeros = 0;
    HistAcc acc, acc2;
    int r0 = blockIdx.x * prep_rows;
    float mf = (float)minFlux, af = (float)addFlux;
    unsigned nzpos = 0; // bit i: this lane met a non-zero value at its i-th column position (any of the rows)
    if (MODE >= 0) {
        // float rows, four pixels per lane; the loads of four rows are issued together (64 B per lane in flight: the kernel
        // lives on memory latency otherwise)
        const float *fs = (const float *)src + (size_t)g * N;
        // (rows are taken four at a time; a group of four whole rows -- all but a frame's last group -- runs without the per-row
        // bounds tests: the sweep issues about as many scalar as vector instructions and both count)
        auto rows4 = [&](auto full_tag, int rb) {
            constexpr bool FULL = decltype(full_tag)::value;
            for (int x4 = threadIdx.x, i = 0; x4 < (w >> 2); x4 += 256, i++) {
                float4 v[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    int r = rb + k;
                    if (FULL || (k < prep_rows && r < h)) v[k] = ((const float4 *)(fs + (size_t)(flip ? (h - 1 - r) : r) * w))[x4];
                }
                if constexpr (RS) { // blotted pixels are +0.0f (v_bfe_i32: all ones where the pixel's bit is set; v_bfi_b32 clears those)
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        int r = rb + k;
                        if (!FULL && (k >= prep_rows || r >= h)) break;
                        const uint32_t nib = rsm[(r - r0) * rsw + (x4 >> 3)] >> ((x4 & 7) * 4);
                        v[k].x = __uint_as_float(__float_as_uint(v[k].x) & ~(uint32_t)(((int)(nib << 31)) >> 31));
                        v[k].y = __uint_as_float(__float_as_uint(v[k].y) & ~(uint32_t)(((int)(nib << 30)) >> 31));
                        v[k].z = __uint_as_float(__float_as_uint(v[k].z) & ~(uint32_t)(((int)(nib << 29)) >> 31));
                        v[k].w = __uint_as_float(__float_as_uint(v[k].w) & ~(uint32_t)(((int)(nib << 28)) >> 31));
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    int r = rb + k;
                    if (!FULL && (k >= prep_rows || r >= h)) break;
                    constexpr int M = MODE >= 0 ? MODE : 0;
                    if constexpr (DELTA && MFPOS && M == 1) {
                        // Sky: when every pixel of the wave's 256 is below 0.5 (NaN compares false) the bright values are all 0
                        // (x <= 0.5 rounds to 0, negatives are clamped) and the dim value of a pixel is 1 where x >= mf2 and 0
                        // elsewhere: with fl(mf2 + af2) > 0.5 (sky_fast, checked on the host) and af2 <= 0.999 (the DELTA
                        // precondition) mf2 <= x < 0.5 gives 0.5 < fl(x + af2) < 1.5, which rounds to 1.  Nine in ten wave-rows of
                        // a sky frame take this branch: four compares instead of two full conversions per pixel.
                        if (sky_fast) {
                            const float4 q = v[k];
                            const bool sky = (q.x < 0.5f) & (q.y < 0.5f) & (q.z < 0.5f) & (q.w < 0.5f);
                            if (__ballot(!sky) == 0ull) {
                                ((uint32_t *)(gout + (size_t)r * w))[x4] = 0u;
                                if (!DELTA && fullbits && lfd_lane() == 0) fullbits[((size_t)g * h + r) * ((w + 255) >> 8) + (x4 >> 6)] = 0ull;
                                acc.n01 += 1;
                                const uint32_t nib = (q.x >= mf2 ? 1u : 0u) | (q.y >= mf2 ? 2u : 0u) | (q.z >= mf2 ? 4u : 0u) | (q.w >= mf2 ? 8u : 0u);
                                acc2.n01 += 1;
                                acc2.ones01 += __popc(nib);
                                uint32_t bw = nib << (4 * (threadIdx.x & 7));
                                bw |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)bw, 0x111, 0xf, 0xf, true); // row_shr:1
                                bw |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)bw, 0x112, 0xf, 0xf, true); // row_shr:2
                                bw |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)bw, 0x114, 0xf, 0xf, true); // row_shr:4
                                if ((threadIdx.x & 7) == 7) { // dim value = bright value (0) + bit, and non-zero exactly where the bit is set
                                    const size_t o = ((size_t)g * h + r) * LFD_WQ(w);
                                    ((uint32_t *)(dbits + o))[x4 >> 3] = bw;
                                    ((uint32_t *)(nzd + o))[x4 >> 3] = bw;
                                }
                                continue;
                            }
                        }
                    }
                    const uint32_t word = prep_word_m<M>(v[k], mf, af);
                    ((uint32_t *)(gout + (size_t)r * w))[x4] = word;
                    if (word) nzpos |= 1u << i;
                    // one bit per aligned word of four pixels: "all four non-zero".  Only where such words line up can a wide
                    // erosion leave anything (k_morph_rect_v decides from these bits without loading the image); a wave's 64
                    // lanes are 64 consecutive words, so the ballot is one u64 of the plane
                    if (!DELTA && fullbits) {
                        u64 fb = __ballot(no_zero_byte(word));
                        if (lfd_lane() == 0) fullbits[((size_t)g * h + r) * ((w + 255) >> 8) + (x4 >> 6)] = fb;
                    }
                    hist_word(word, sh[wv], acc);
                    if (DELTA) {
                        // (minFlux > 0, the usual case: x < mf covers x < 0, and x >= mf implies x > 0: one compare, one select)
                        auto dimv = [&](float x) -> float {
                            if constexpr (MFPOS) return x < mf2 ? 0.0f : __fadd_rn(x, af2);
                            else return prep_mask_m<3>(x, mf2, af2);
                        };
                        const uint32_t wd = pack_sat_u8x4(dimv(v[k].x), dimv(v[k].y), dimv(v[k].z), dimv(v[k].w));
                        hist_word(wd, sh[4 + wv], acc2);
                        const uint32_t df = wd - word; // 0 or 1 per byte (see above): no borrows
                        uint32_t nib = lsb_nibble(df);
                        // eight lanes (32 pixels) make one 32-bit half of a bit-row word: OR over the row of lanes (DPP row_shr)
                        uint32_t bw = nib << (4 * (threadIdx.x & 7));
                        bw |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)bw, 0x111, 0xf, 0xf, true); // row_shr:1
                        bw |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)bw, 0x112, 0xf, 0xf, true); // row_shr:2
                        bw |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)bw, 0x114, 0xf, 0xf, true); // row_shr:4
                        if ((threadIdx.x & 7) == 7) ((uint32_t *)(dbits + ((size_t)g * h + r) * LFD_WQ(w)))[x4 >> 3] = bw;
                        // ... and the same for "dim value non-zero" (k_bits_erode decides from these bits alone where an erosion
                        // can leave anything)
                        const uint32_t y_ = (((wd | 0x80808080u) - 0x01010101u) | wd) & 0x80808080u; // 0x80 per non-zero byte
                        uint32_t nw = lsb_nibble(y_ >> 7);
                        nw <<= 4 * (threadIdx.x & 7);
                        nw |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)nw, 0x111, 0xf, 0xf, true);
                        nw |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)nw, 0x112, 0xf, 0xf, true);
                        nw |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)nw, 0x114, 0xf, 0xf, true);
                        if ((threadIdx.x & 7) == 7) ((uint32_t *)(nzd + ((size_t)g * h + r) * LFD_WQ(w)))[x4 >> 3] = nw;
                    }
                }
            }
        };
        for (int rb = r0; rb < r0 + prep_rows && rb < h; rb += 4) {
            if (rb + 4 <= r0 + prep_rows && rb + 4 <= h) rows4(std::true_type{}, rb);
            else rows4(std::false_type{}, rb);
        }
    } else
    for (int r = r0; r < r0 + prep_rows && r < h; r++) {
        int sr = flip ? (h - 1 - r) : r;
        {
            for (int x = threadIdx.x, i = 0; x < w; x += 256, i++) {
                unsigned a;
                size_t k = (size_t)g * N + (size_t)sr * w + x;
                if (dtype == 0) a = ((const uint8_t *)src)[k];
                else if (dtype == 1) a = prep_f32(((const float *)src)[k], mode, mf, af);
                else a = prep_f64(((const double *)src)[k], mode, minFlux, addFlux);
                gout[(size_t)r * w + x] = (uint8_t)a;
                if (a) nzpos |= 1u << i;
                if (a) atomicAdd(&sh[wv][a], 1); else zeros++;
            }
        }
    }
    // zeros dominate sky frames: count them in registers, one LDS add per wave
    acc.zeros += zeros;
    hist_flush(acc, sh[wv]);
    if (DELTA) hist_flush(acc2, sh[4 + wv]);
    // occupied cells of this workgroup's rows (one band): position i of wave wv covers ppl pixels per lane
    // from column 256 ppl i + 64 ppl wv on, i.e. (64 ppl / 16) cells
    if (cellbm && r0 < h) {
        const int ppl = MODE >= 0 ? 4 : 1, lpc = CELLBM_COLS / ppl; // lanes per cell: 4 or 16
        u64 *bw = cellbm + ((size_t)g * bm_bands + r0 / CELLBM_ROWS) * CELLBM_WORDS;
        for (int i = 0; (256 * ppl) * i < w; i++) {
            u64 m = __ballot((nzpos >> i) & 1u);
            if (!m || lfd_lane() != 0) continue;
            u64 cells = 0;
            for (int k = 0; k < 64 / lpc; k++)
                if ((m >> (k * lpc)) & (lpc == 4 ? 0xFull : 0xFFFFull)) cells |= 1ull << k;
            int c0 = (256 * ppl * i + 64 * ppl * wv) / CELLBM_COLS; // first cell of this wave's stretch (16 or 4 cells)
            atomicOr((unsigned long long *)&bw[c0 >> 6], cells << (c0 & 63));
        }
    }
    __syncthreads();
    int b = threadIdx.x;
    int s = sh[0][b] + sh[1][b] + sh[2][b] + sh[3][b];
    if (s) atomicAdd(&hist[g * 256 + b], s);
    if (DELTA) {
        int s2 = sh[4][b] + sh[5][b] + sh[6][b] + sh[7][b];
        if (s2) atomicAdd(&hist2[g * 256 + b], s2);
    }
}

// ------------------------------------------------------------------------------------------
// prep + histogram + erosion in one pass (the dim pass's front end: processfield.py:453-464).
// A workgroup owns a band of BR image rows over the full width: the float rows of the band and
// of the (kh-1) halo rows are converted once into an LDS band (16-byte 0xFF pads left and right,
// 0xFF rows outside the image: an erosion ignores what lies outside), the histogram is taken
// from the band's own rows (equalizeHist sees the un-eroded image; its monotone LUT is applied
// after the morphology), then the separable minimum runs on the LDS band four pixels per lane
// (even / odd bytes as packed u16, v_pk_min_u16) and the eroded rows are stored 16 B per lane.
// The 8-bit image is never written or re-read: 4N(1 + (kh-1)/BR) + 1N bytes instead of 5N + 2N.
// float32 input, all-ones kernel, w % 16 == 0, kw/2 <= 16 and kw - 1 - kw/2 <= 16.
// ------------------------------------------------------------------------------------------
#define PE_THREADS 1024
#define PE_HISTS 16 // private LDS histograms (one per wave; the DUAL variant gives eight to each of its two outputs)
// DUAL: the same pass over the float rows also produces the BRIGHT pass's 8-bit image (mode 1: x < 0 -> 0), its histogram
// and its cell occupancy (gray_b / hist_b / cellbm_b) for the band's own rows: lfdmi_detect_batch runs both passes on the
// same frames, so the 12.2 MB of a frame cross HBM once instead of twice (the dim outputs of frames the bright pass then
// accepts are not used).
template <bool DUAL, int MODE>
__global__ void __launch_bounds__(PE_THREADS)
k_prep_erode(const float *src, int h, int w, int flip, int mode, float mf, float af, uint8_t *dst, int *hist, int kh, int kw,
             int BR, u64 *cellbm, int bm_bands, const int *active, uint8_t *gray_b, int *hist_b, u64 *cellbm_b) {
    int g = blockIdx.y;
    if (active && !active[g]) return;
    extern __shared__ __attribute__((aligned(16))) uint8_t smb[];
    __shared__ int sh[PE_HISTS][256];
    constexpr int NH = DUAL ? PE_HISTS / 2 : PE_HISTS; // rows 0 .. NH-1: this kernel's own image; NH .. : the bright image
    const int R = BR + kh - 1, S = w + 32, SW = S >> 2, W4 = w >> 2;
    uint32_t *band = (uint32_t *)smb;            // R x S bytes
    uint32_t *vbuf = band + R * SW;              // BR x S bytes: vertical minimum
    for (int k = threadIdx.x; k < PE_HISTS * 256; k += PE_THREADS) ((int *)sh)[k] = 0;
    const int y0 = blockIdx.x * BR, ay = kh / 2, ax = kw / 2;
    const size_t N = (size_t)h * w;
    const float *s = src + (size_t)g * N;
    const int wv = (threadIdx.x >> 6) & (NH - 1);
    // pads of every staged row
    for (int it = threadIdx.x; it < R * 8; it += PE_THREADS) {
        int r = it >> 3, k = it & 7;
        band[r * SW + (k < 4 ? k : W4 + k)] = 0xFFFFFFFFu;
    }
    __syncthreads();
    HistAcc acc, acc_b;
    // piece `it` of the band is (row it / W4, 16-byte column it % W4); a lane's pieces are PE_THREADS apart, so it carries
    // (row, column) along instead of dividing for every piece (an integer division is ~40 vector instructions)
    const int dr1 = PE_THREADS / W4, dx1 = PE_THREADS - dr1 * W4;
    int r_c = threadIdx.x / W4, x_c = threadIdx.x - r_c * W4;
    for (int it0 = threadIdx.x; it0 < R * W4; it0 += 4 * PE_THREADS) { // four row pieces in flight per lane
        float4 v[4];
        int gyv[4], rr[4], xx[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            rr[u] = r_c; xx[u] = x_c;
            x_c += dx1; r_c += dr1;
            if (x_c >= W4) { x_c -= W4; r_c++; }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int r = rr[u], x4 = xx[u];
            int gy = y0 - ay + r;
            gyv[u] = (r < R && gy >= 0 && gy < h) ? gy : -1;
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gyv[u] >= 0) v[u] = ((const float4 *)(s + (size_t)(flip ? (h - 1 - gy) : gy) * w))[x4];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int r = rr[u], x4 = xx[u];
            if (r >= R) continue;
            uint32_t word = 0xFFFFFFFFu;
            if (gyv[u] >= 0) {
                word = prep_word_m<MODE>(v[u], mf, af);
                if (gyv[u] >= y0 && gyv[u] < y0 + BR) { // the band's own rows: every image row is counted once
                    hist_word(word, sh[wv], acc);
                    if (DUAL) { // the bright pass's image of the same pixels
                        const uint32_t word2 = prep_word_m<1>(v[u], 0.f, 0.f);
                        ((uint32_t *)(gray_b + (size_t)g * N + (size_t)gyv[u] * w))[x4] = word2;
                        hist_word(word2, sh[NH + wv], acc_b);
                        if (word2) { // (the bright image is sparse: a few thousand marks per frame)
                            int cx = x4 >> 2;
                            atomicOr((unsigned long long *)&cellbm_b[((size_t)g * bm_bands + gyv[u] / CELLBM_ROWS) * CELLBM_WORDS + (cx >> 6)], 1ull << (cx & 63));
                        }
                    }
                }
            }
            band[r * SW + 4 + x4] = word;
        }
    }
    hist_flush(acc, sh[wv]);
    if (DUAL) hist_flush(acc_b, sh[NH + wv]);
    __syncthreads();
    if (threadIdx.x < 256) {
        int b = threadIdx.x, t = 0;
        for (int k = 0; k < NH; k++) t += sh[k][b];
        if (t) atomicAdd(&hist[g * 256 + b], t);
    } else if (DUAL && threadIdx.x < 512) {
        int b = threadIdx.x - 256, t = 0;
        for (int k = 0; k < NH; k++) t += sh[NH + k][b];
        if (t) atomicAdd(&hist_b[g * 256 + b], t);
    }
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    // vertical minimum, pads included (they stay 0xFF)
    const int dro = PE_THREADS / SW, djo = PE_THREADS - dro * SW;
    int o_c = threadIdx.x / SW, j_c = threadIdx.x - o_c * SW;
    for (int it = threadIdx.x; it < BR * SW; it += PE_THREADS) {
        const int o = o_c, j = j_c;
        j_c += djo; o_c += dro;
        if (j_c >= SW) { j_c -= SW; o_c++; }
        us2 mE = __builtin_bit_cast(us2, 0x00FF00FFu), mO = mE;
        for (int dy = 0; dy < kh; dy++) {
            uint32_t t = band[(o + dy) * SW + j];
            mE = __builtin_elementwise_min(mE, __builtin_bit_cast(us2, t & 0x00FF00FFu));
            mO = __builtin_elementwise_min(mO, __builtin_bit_cast(us2, (t >> 8) & 0x00FF00FFu));
        }
        vbuf[it] = __builtin_bit_cast(uint32_t, mE) | (__builtin_bit_cast(uint32_t, mO) << 8);
    }
    __syncthreads();
    // horizontal minimum and store: 16 output bytes per lane
    uint8_t *d = dst + (size_t)g * N;
    const int W16 = w >> 4;
    const bool near_only = ax <= 4 && kw - 1 - ax <= 4; // the window reaches at most one word beyond the piece
    const int drh = PE_THREADS / W16, dxh = PE_THREADS - drh * W16;
    int oh_c = threadIdx.x / W16, xh_c = threadIdx.x - oh_c * W16;
    for (int it = threadIdx.x; it < BR * W16; it += PE_THREADS) {
        const int o = oh_c, x16 = xh_c;
        xh_c += dxh; oh_c += drh;
        if (xh_c >= W16) { xh_c -= W16; oh_c++; }
        int gy = y0 + o;
        if (gy >= h) continue;
        const uint32_t *rw = vbuf + o * SW;
        uint32_t outw[4];
        if (near_only) {
            // the vertical minimum of a sky frame is zero almost everywhere (a byte survives only where kh rows are all
            // non-zero): a piece whose 16 bytes and the words either side of them are clear erodes to zeros
            const uint4 cpc = *(const uint4 *)(rw + 4 + 4 * x16);
            if ((cpc.x | cpc.y | cpc.z | cpc.w | rw[3 + 4 * x16] | rw[8 + 4 * x16]) == 0u) {
                *(uint4 *)(d + (size_t)gy * w + 16 * x16) = make_uint4(0, 0, 0, 0);
                continue;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            int ob = 16 + 16 * x16 + 4 * q - ax; // byte offset of the window's first column in the padded row
            int qi = ob >> 2;
            uint32_t lo = rw[qi], hi = rw[qi + 1];
            us2 mE = __builtin_bit_cast(us2, 0x00FF00FFu), mO = mE;
            for (int dx = 0; dx < kw; dx++, ob++) {
                if ((ob >> 2) != qi) { qi = ob >> 2; lo = hi; hi = rw[qi + 1]; }
                uint32_t sft = __builtin_amdgcn_alignbyte(hi, lo, (unsigned)(ob & 3));
                mE = __builtin_elementwise_min(mE, __builtin_bit_cast(us2, sft & 0x00FF00FFu));
                mO = __builtin_elementwise_min(mO, __builtin_bit_cast(us2, (sft >> 8) & 0x00FF00FFu));
            }
            outw[q] = __builtin_bit_cast(uint32_t, mE) | (__builtin_bit_cast(uint32_t, mO) << 8);
        }
        *(uint4 *)(d + (size_t)gy * w + 16 * x16) = make_uint4(outw[0], outw[1], outw[2], outw[3]);
        // a lane's 16 bytes are one cell of the occupancy bitmap
        if (cellbm && (outw[0] | outw[1] | outw[2] | outw[3]))
            atomicOr((unsigned long long *)&cellbm[((size_t)g * bm_bands + gy / CELLBM_ROWS) * CELLBM_WORDS + (x16 >> 6)], 1ull << (x16 & 63));
    }
}

// Erosion (all-ones kh x kw, anchor at the centre, outside pixels ignored) of the DIM pass's 8-bit image without that image
// ever being stored: k_prep_hist<1, true> left the bright image (gsrc), one bit per pixel "dim value = bright value + 1"
// (dbits) and one bit per pixel "dim value non-zero" (nzd).  An eroded pixel is non-zero exactly where all kh x kw input
// pixels are, which is bit arithmetic on the nzd rows: a lane owns one 64-pixel word of a row, ANDs the shifted words of the
// kh rows, and the wave writes zeros for its 4096 pixels with 16-byte stores.  Only for the surviving bits (objects: the
// sky is zeros and ones at random, P(3 x 3 all non-zero) ~ 1e-6) are the window's values fetched and the minimum stored.
// The float frames are not read again and the sky costs a few bit operations per 64 pixels.
#define BE_REP 1 // groups of 64 words per wave (4 measured 0.33 ms against 0.27: the zero fill wants as many waves in flight as it can get)
__global__ void __launch_bounds__(256)
k_bits_erode(const uint8_t *gsrc, const u64 *dbits, const u64 *nzd, uint8_t *dst, u64 *cellbm, int bm_bands, int h, int w, int kh, int kw,
             const int *active) {
    const int g = blockIdx.y;
    if (active && !active[g]) return;
    const int wq = LFD_WQ(w), nw = h * wq, lane = threadIdx.x & 63;
    for (int rep = 0; rep < BE_REP; rep++) {
    const int i0 = ((blockIdx.x * 4 + (threadIdx.x >> 6)) * BE_REP + rep) * 64; // this group's 64 words
    if (i0 >= nw) return;
    const int i = i0 + lane;
    const int ay = kh / 2, ax = kw / 2;
    const size_t N = (size_t)h * w;
    const u64 *nz = nzd + (size_t)g * nw;
    u64 e = 0ull;
    int y = 0, q = 0;
    if (i < nw) { y = i / wq; q = i - y * wq; }
    // zeros for the wave's pixels FIRST (64 words = 4096 bytes, contiguous: rows are whole words when w % 64 == 0; otherwise a
    // word's bytes beyond the row end belong to nobody and are skipped): the stores depend on nothing, so they leave before the
    // wave waits for its bit rows; the survivors' values follow in program order
    uint8_t *d = dst + (size_t)g * N;
    if ((w & 63) == 0) {
        uint4 *dz = (uint4 *)(d + (size_t)i0 * 64);
        const int n16 = min(64, nw - i0) * 4;
        for (int k = lane; k < n16; k += 64) dz[k] = make_uint4(0, 0, 0, 0);
    } else if (i < nw) {
        const int x0 = q << 6, nb = min(64, w - x0);
        for (int b = 0; b < nb; b++) d[(size_t)y * w + x0 + b] = 0; // (odd widths are not on the batch path: plain and slow)
    }
    if (i < nw) {
        e = valid_mask(q, w);
        for (int dy = 0; dy < kh; dy++) {
            const int yy = y + dy - ay;
            if (yy < 0 || yy >= h) continue; // rows outside the image do not constrain
            const u64 *row = nz + (size_t)yy * wq;
            const u64 c = row[q] | ~valid_mask(q, w);           // columns outside the image do not constrain either
            const u64 p = q > 0 ? row[q - 1] : ~0ull;
            const u64 n = q + 1 < wq ? (row[q + 1] | ~valid_mask(q + 1, w)) : ~0ull;
            u64 ha = c;
            for (int sft = 1; sft <= kw - 1 - ax; sft++) ha &= (c >> sft) | (n << (64 - sft)); // columns x + sft
            for (int sft = 1; sft <= ax; sft++) ha &= (c << sft) | (p >> (64 - sft));           // columns x - sft
            e &= ha;
        }
    }
    u64 todo = __ballot(e != 0ull);
    if (todo == 0ull) continue;
    if (e) { // cells of the output (16 x 16 pixels: four per word)
        u64 cells = 0ull;
        for (int c = 0; c < 4; c++)
            if ((e >> (16 * c)) & 0xFFFFull) cells |= 1ull << c;
        const int cx0 = q * 4;
        atomicOr((unsigned long long *)&cellbm[((size_t)g * bm_bands + y / CELLBM_ROWS) * CELLBM_WORDS + (cx0 >> 6)], cells << (cx0 & 63));
    }
    const uint8_t *gs = gsrc + (size_t)g * N;
    const u64 *db = dbits + (size_t)g * nw;
    // The survivors of the wave's 64 words are spread over the lanes: every lane queues up to eight of its word's bits per
    // round in a list in LDS (position from a prefix sum over the lanes), then the wave works through the list a pixel per
    // lane (a word per iteration kept one lane in eight busy: survivors come as short stretches across a trail or a star).
    __shared__ unsigned short qlist[4][512];
    __shared__ int qyq[4][64];
    const int wvi = threadIdx.x >> 6;
    qyq[wvi][lane] = (y << 8) | q; // q < 128 (image widths up to 8191)
    u64 rem = e;
    for (;;) {
        const int cnt = min(__popcll(rem), 8);
        int incl = cnt;
        for (int off = 1; off < 64; off <<= 1) {
            int t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        const int total = __shfl(incl, 63);
        if (total == 0) break;
        int o = incl - cnt;
        for (int k = 0; k < cnt; k++) {
            const int bb = __ffsll((long long)rem) - 1;
            rem &= rem - 1;
            qlist[wvi][o++] = (unsigned short)((lane << 6) | bb);
        }
        __builtin_amdgcn_wave_barrier();
        for (int t = lane; t < total; t += 64) {
            const unsigned ent = qlist[wvi][t];
            const int yq = qyq[wvi][ent >> 6];
            const int yy0 = yq >> 8, x = ((yq & 0xff) << 6) + (int)(ent & 63u);
            auto val = [&](int yy, int xx) -> unsigned {
                if (yy < 0 || yy >= h || xx < 0 || xx >= w) return 255u; // outside pixels are ignored by an erosion
                return (unsigned)gs[(size_t)yy * w + xx] + (unsigned)((db[(size_t)yy * wq + (xx >> 6)] >> (xx & 63)) & 1ull);
            };
            unsigned m = 255u;
            if (kh == 3 && kw == 3 && x >= 1 && x + 2 < w && yy0 >= 1 && yy0 + 1 < h) {
                // the reference's default, away from the frame's border: a row's three values are one (unaligned) 4-byte load of the
                // bright image and one word of the bit plane (two when the three bits straddle a word): 6 - 9 loads in flight
                // instead of 18.  This kernel's time is the gathers of the survivors' windows (one CU retires a 64-lane gather at
                // ~1.5 clocks per lane), not its zero fill (without it: 0.195 -> 0.180 ms) and not its bit rows
                unsigned gv[3];
                u64 bw0[3], bw1[3];
                const int xb = x - 1, wi = xb >> 6, sh = xb & 63;
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const size_t ro = (size_t)(yy0 + k - 1);
                    uint32_t t4;
                    __builtin_memcpy(&t4, gs + ro * w + xb, 4);
                    gv[k] = t4;
                    bw0[k] = db[ro * wq + wi];
                    bw1[k] = sh > 61 ? db[ro * wq + wi + 1] : 0ull; // (x + 1 <= w - 2: that word exists)
                }
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const unsigned bits = (unsigned)((bw0[k] >> sh) | (sh > 61 ? bw1[k] << (64 - sh) : 0ull)) & 7u;
                    m = min(m, (gv[k] & 0xffu) + (bits & 1u));
                    m = min(m, ((gv[k] >> 8) & 0xffu) + ((bits >> 1) & 1u));
                    m = min(m, ((gv[k] >> 16) & 0xffu) + ((bits >> 2) & 1u));
                }
            } else if (kh == 3 && kw == 3) { // (border pixels: outside values are ignored)
                unsigned v9[9];
#pragma unroll
                for (int k = 0; k < 9; k++) v9[k] = val(yy0 + k / 3 - 1, x + k % 3 - 1);
#pragma unroll
                for (int k = 0; k < 9; k++) m = min(m, v9[k]);
            } else {
                for (int dy = 0; dy < kh; dy++)
                    for (int dx = 0; dx < kw; dx++) m = min(m, val(yy0 + dy - ay, x + dx - ax));
            }
            d[(size_t)yy0 * w + x] = (uint8_t)m;
        }
        __builtin_amdgcn_wave_barrier();
    }
    } // rep
}

// histogram only (for lfdmi_equalize_hist on an existing u8 image)
__global__ void __launch_bounds__(256)
k_hist_u8(const uint8_t *src, size_t N, int *hist) {
    int g = blockIdx.y;
    __shared__ int sh[256];
    sh[threadIdx.x] = 0;
    __syncthreads();
    const uint8_t *s = src + (size_t)g * N;
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < N; k += (size_t)gridDim.x * 256)
        atomicAdd(&sh[s[k]], 1);
    __syncthreads();
    if (sh[threadIdx.x]) atomicAdd(&hist[g * 256 + threadIdx.x], sh[threadIdx.x]);
}

// ------------------------------------------------------------------------------------------
// equalizeHist LUT (OpenCV: first non-empty bin i, scale = 255.f/(total-hist[i]),
// lut[b] = saturate(round(float(sum_{i<k<=b} hist[k]) * scale))).  One workgroup per frame.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_lut(const int *hist, int total, uint8_t *lut, const int *active) {
    int g = blockIdx.x, t = threadIdx.x;
    if (active && !active[g]) return;
    __shared__ int cum[256];
    __shared__ int first;
    int hv = hist[g * 256 + t];
    cum[t] = hv;
    if (t == 0) first = 256;
    __syncthreads();
    if (hv) atomicMin(&first, t);
    for (int off = 1; off < 256; off <<= 1) {
        int v = (t >= off) ? cum[t - off] : 0;
        __syncthreads();
        cum[t] += v;
        __syncthreads();
    }
    int f = first;
    if (f >= 256) { lut[g * 256 + t] = 0; return; }
    int hf = hist[g * 256 + f];
    unsigned out;
    if (hf == total) out = (unsigned)f;
    else if (t <= f) out = 0;
    else {
        float scale = __fdiv_rn(255.f, (float)(total - hf));
        out = sat_u8_f32(__fmul_rn((float)(cum[t] - cum[f]), scale));
    }
    lut[g * 256 + t] = (uint8_t)out;
}

__global__ void __launch_bounds__(256)
k_apply_lut(const uint8_t *src, const uint8_t *lut, uint8_t *dst, size_t N) {
    int g = blockIdx.y;
    __shared__ uint8_t l[256];
    l[threadIdx.x] = lut[g * 256 + threadIdx.x];
    __syncthreads();
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < N; k += (size_t)gridDim.x * 256)
        dst[(size_t)g * N + k] = l[src[(size_t)g * N + k]];
}

// ---- SWAR helpers of the wave kernels: four pixels as two packed-u16 pairs (E = bytes 0, 2; O = bytes 1, 3) ----
typedef unsigned short dcw_us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t dcw_pkmax(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(dcw_us2, a), __builtin_bit_cast(dcw_us2, b)));
}
__device__ __forceinline__ uint32_t dcw_pkmin(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(dcw_us2, a), __builtin_bit_cast(dcw_us2, b)));
}
// (E, O)[i] <- max((E, O)[i], the same bytes S positions further right), for words 0 .. nw-1 of a split byte row.
// Byte x + S of a split row: S = 4k: word i + k; S = 4k + 1: E' = O[i+k], O' = (E[i+k] >> 16 | E[i+k+1] << 16); ...
template <int S, int NWT, bool MIN = false>
__device__ __forceinline__ void dcw_shift_max(uint32_t (&E)[NWT], uint32_t (&O)[NWT], int nw) {
    constexpr int K = S / 4, R = S % 4;
#pragma unroll
    for (int i = 0; i < NWT; i++) {
        if (i >= nw || i + K + 1 >= NWT) continue;
        uint32_t e2, o2;
        if constexpr (R == 0) { e2 = E[i + K]; o2 = O[i + K]; }
        else if constexpr (R == 1) { e2 = O[i + K]; o2 = __builtin_amdgcn_alignbyte(E[i + K + 1], E[i + K], 2); }
        else if constexpr (R == 2) { e2 = __builtin_amdgcn_alignbyte(E[i + K + 1], E[i + K], 2); o2 = __builtin_amdgcn_alignbyte(O[i + K + 1], O[i + K], 2); }
        else { e2 = __builtin_amdgcn_alignbyte(O[i + K + 1], O[i + K], 2); o2 = E[i + K + 1]; }
        E[i] = MIN ? dcw_pkmin(E[i], e2) : dcw_pkmax(E[i], e2);
        O[i] = MIN ? dcw_pkmin(O[i], o2) : dcw_pkmax(O[i], o2);
    }
}
// max over the KW bytes starting at each of the four pixels of output word jj of a staged row (the window of pixel q
// starts at staged byte CANNY_MOFF - KW/2 + 4 jj + q): running maxima over windows of 1, 2, 4, ... bytes, then
// max(m_p[x], m_p[x + KW - p]) for the largest power of two p <= KW -- ~3 packed instructions per doubling step and
// word instead of ~6 per tap.
// (MIN: running minima for an erosion; BASE: staged byte of output word 0's first pixel -- CANNY_MOFF in the fused tile
// kernels, MORPH_HALO in k_morph_rect_v.)  Bytes beyond the window that the shifts drag in are zero for a maximum and
// never reach word 0 of the result in either case.
template <int KW, bool MIN = false, int BASE = 12>
__device__ __forceinline__ uint32_t dcw_hmax(const uint32_t *rw, int jj) {
    constexpr int AX = KW / 2, R0 = (BASE - AX) & 3, Q0 = (BASE - AX) >> 2;
    constexpr int NB = KW + 3, NW = (NB + 3) / 4, NR = (R0 + NB + 3) / 4, NWT = NW + 9;
    uint32_t raw[NR + 1];
#pragma unroll
    for (int i = 0; i < NR; i++) raw[i] = rw[Q0 + jj + i];
    raw[NR] = 0;
    uint32_t E[NWT], O[NWT];
#pragma unroll
    for (int i = 0; i < NWT; i++) {
        uint32_t a = 0;
        if (i < NW) a = R0 ? __builtin_amdgcn_alignbyte(raw[i + 1 <= NR ? i + 1 : NR], raw[i], R0) : raw[i];
        E[i] = a & 0x00FF00FFu;
        O[i] = (a >> 8) & 0x00FF00FFu;
    }
    constexpr int P = KW >= 16 ? 16 : (KW >= 8 ? 8 : (KW >= 4 ? 4 : (KW >= 2 ? 2 : 1))), D = KW - P;
    // words each step still has to produce (bytes needed by the steps after it)
    constexpr int need1 = 4 + D + (P > 8 ? 8 : 0) + (P > 4 ? 4 : 0) + (P > 2 ? 2 : 0);
    if constexpr (P >= 2) dcw_shift_max<1, NWT, MIN>(E, O, (need1 + 3) / 4);
    constexpr int need2 = need1 - 2;
    if constexpr (P >= 4) dcw_shift_max<2, NWT, MIN>(E, O, (need2 + 3) / 4);
    constexpr int need4 = need2 - 4;
    if constexpr (P >= 8) dcw_shift_max<4, NWT, MIN>(E, O, (need4 + 3) / 4);
    if constexpr (P >= 16) dcw_shift_max<8, NWT, MIN>(E, O, (4 + D + 3) / 4);
    if constexpr (D > 0) dcw_shift_max<D, NWT, MIN>(E, O, 1);
    return E[0] | (O[0] << 8);
}

// ------------------------------------------------------------------------------------------
// erode / dilate with an all-ones kh x kw kernel: separable running min/max staged in LDS.
// dst(y,x) = op over dy<kh, dx<kw of src(y+dy-kh/2, x+dx-kw/2), out-of-image samples ignored.
// Optional monotone LUT applied on the way out (equalizeHist commutes with min/max because
// the LUT is non-decreasing), and optional "!= 0" bit-mask for the Hough pixel list.
// ------------------------------------------------------------------------------------------
#define MORPH_TW 64
#define MORPH_TH 32

template <int OP> // 0 dilate (max), 1 erode (min)
__global__ void __launch_bounds__(256)
k_morph_rect(const uint8_t *src, uint8_t *dst, u64 *bits, const uint8_t *lut, int h, int w, int kh,
             int kw, const int *active) {
    int g = blockIdx.z;
    if (active && !active[g]) return;
    extern __shared__ uint8_t sm[];
    const int IH = MORPH_TH + kh - 1, IW = MORPH_TW + kw - 1;
    uint8_t *tin = sm;                     // IH x IW
    uint8_t *tmp = sm + ((IH * IW + 15) & ~15); // IH x MORPH_TW
    __shared__ uint8_t slut[256];
    const int fill = OP ? 255 : 0;
    int x0 = blockIdx.x * MORPH_TW, y0 = blockIdx.y * MORPH_TH;
    int ay = kh / 2, ax = kw / 2;
    size_t N = (size_t)h * w;
    const uint8_t *s = src + (size_t)g * N;
    if (lut) slut[threadIdx.x] = lut[g * 256 + threadIdx.x];
    for (int idx = threadIdx.x; idx < IH * IW; idx += 256) {
        int iy = idx / IW, ix = idx - iy * IW;
        int gy = y0 + iy - ay, gx = x0 + ix - ax;
        int v = fill;
        if (gy >= 0 && gy < h && gx >= 0 && gx < w) v = s[(size_t)gy * w + gx];
        tin[idx] = (uint8_t)v;
    }
    __syncthreads();
    int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int ry = wv; ry < IH; ry += 4) {
        int m = fill;
        const uint8_t *row = tin + ry * IW + lane;
        for (int dx = 0; dx < kw; dx++) {
            int v = row[dx];
            m = OP ? min(m, v) : max(m, v);
        }
        tmp[ry * MORPH_TW + lane] = (uint8_t)m;
    }
    __syncthreads();
    int wq = LFD_WQ(w);
    for (int oy = wv; oy < MORPH_TH; oy += 4) {
        int m = fill;
        for (int dy = 0; dy < kh; dy++) {
            int v = tmp[(oy + dy) * MORPH_TW + lane];
            m = OP ? min(m, v) : max(m, v);
        }
        int gy = y0 + oy, gx = x0 + lane;
        bool valid = gy < h && gx < w;
        int out = lut ? slut[m] : m;
        if (valid) dst[(size_t)g * N + (size_t)gy * w + gx] = (uint8_t)out;
        if (bits) {
            u64 bal = __ballot(valid && out != 0);
            if (lane == 0 && gy < h) bits[(size_t)g * h * wq + (size_t)gy * wq + blockIdx.x] = bal;
        }
    }
}

// Same operator for frames whose width is a multiple of 16 (SDSS 2048, 4096): the tile and a
// 16-byte halo are fetched as aligned dwords, an all-zero input tile (sky after the bright clamp,
// or after erosion in the dim pass) short-circuits a dilation, and the result tile is staged in
// LDS and written with 16 B per lane.
#define MORPH_HALO 16

// KHC, KWC > 0: the kernel's size at compile time (unrolled; the horizontal pass by doubling, dcw_hmax); 0: run-time sizes.
// Erosion (OP == 1) of a sparse image -- the dim pass's 8-bit image is mostly zeros with isolated ones -- skips what cannot
// survive: a kw-wide window of non-zero bytes must contain a whole aligned word of non-zero bytes (kw >= 7), so a staged row
// without such a word erodes to zeros, an output row needs kh live input rows, and a tile without a live row is all zeros
// without touching LDS.
template <int OP, int KHC, int KWC>
__device__ __forceinline__ void morph_rect_tile(const uint8_t *src, uint8_t *dst, u64 *bits, const uint8_t *lut, int h, int w, int kh,
                                                int kw, u64 *cellout, int bm_bands, const int bx, const int by, const int g) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smv[];
    const int IH = MORPH_TH + kh - 1;
    const int IWB = MORPH_TW + 2 * MORPH_HALO; // bytes per staged row
    uint8_t *tin = smv;                       // IH x IWB
    uint8_t *tmp = smv + IH * IWB;            // IH x MORPH_TW x 2 (split even/odd bytes)
    uint8_t *tout = tmp + IH * MORPH_TW * 2;  // MORPH_TH x MORPH_TW
    __shared__ uint8_t slut[256];
    __shared__ int rowflag[MORPH_TH + LFDMI_MAX_MORPH_K]; // dilation: staged row holds a non-zero byte
    const uint32_t fillw = OP ? 0xffffffffu : 0u;
    int x0 = bx * MORPH_TW, y0 = by * MORPH_TH;
    int ay = kh / 2, ax = kw / 2;
    size_t N = (size_t)h * w;
    const uint8_t *s = src + (size_t)g * N;
    if (lut) slut[threadIdx.x] = lut[g * 256 + threadIdx.x];
    if (threadIdx.x < IH) rowflag[threadIdx.x] = 0;
    const bool sparse_erode = OP == 1 && kw >= 7; // (a window of >= 7 bytes always contains an aligned 4-byte word)
    const bool known_empty = false;
    __syncthreads();
    uint32_t any = 0;
    const int IW16 = IWB / 16; // 16-byte pieces per staged row (6)
    for (int idx = threadIdx.x; idx < IH * IW16 && !known_empty; idx += 256) {
        int iy = idx / IW16, wx = idx - iy * IW16;
        int gy = y0 + iy - ay, gx = x0 - MORPH_HALO + 16 * wx;
        uint4 v = make_uint4(fillw, fillw, fillw, fillw);
        if (gy >= 0 && gy < h && gx >= 0 && gx < w) v = *(const uint4 *)(s + (size_t)gy * w + gx);
        uint32_t nzv = v.x | v.y | v.z | v.w;
        if (OP == 1) {
            if (sparse_erode) { // any word without a zero byte?  haszero(x) = (x - 0x01010101) & ~x & 0x80808080
                auto full = [](uint32_t x) { return ((x - 0x01010101u) & ~x & 0x80808080u) == 0u; };
                nzv = (full(v.x) || full(v.y) || full(v.z) || full(v.w)) ? 1u : 0u;
            } else nzv = 1u;
        }
        if (nzv) rowflag[iy] = 1;
        any |= nzv;
        ((uint4 *)tin)[idx] = v;
    }
    int wq = LFD_WQ(w);
    int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int nz = __syncthreads_or(any != 0);
    uint8_t *d = dst + (size_t)g * N;
    if (!nz) {
        // dilation of an all-zero neighbourhood / erosion of rows none of which holds a run of kw non-zero bytes: zeros
        uint32_t z = lut ? slut[0] : 0u;
        z |= z << 8; z |= z << 16;
        if (threadIdx.x < MORPH_TH * 4) {
            int row = threadIdx.x >> 2, c16 = threadIdx.x & 3;
            int gy = y0 + row, gx = x0 + 16 * c16;
            if (gy < h && gx < w) *(uint4 *)(d + (size_t)gy * w + gx) = make_uint4(z, z, z, z);
        }
        if (bits) {
            u64 bal = z ? valid_mask(bx, w) : 0ull;
            if (threadIdx.x < MORPH_TH && y0 + threadIdx.x < h)
                bits[(size_t)g * h * wq + (size_t)(y0 + threadIdx.x) * wq + bx] = bal;
        }
        return;
    }
    // Four pixels per lane: a dword of the tile is split into its even and odd bytes, each as
    // two 16-bit lanes (0x00FF00FF masks), so that one v_pk_max_u16 / v_pk_min_u16 handles two
    // pixels and LDS traffic is dwords, not bytes.  tmp keeps the horizontal result in that split
    // form: (IH x 16) even words followed by (IH x 16) odd words.
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    uint32_t *tinw = (uint32_t *)tin;
    uint32_t *tmpE = (uint32_t *)tmp, *tmpO = tmpE + IH * (MORPH_TW / 4);
    const uint32_t fsplit = OP ? 0x00FF00FFu : 0u;
    for (int it = threadIdx.x; it < IH * (MORPH_TW / 4); it += 256) {
        int ry = it >> 4, j = it & 15;
        uint32_t aE = fsplit, aO = fsplit;
        if (!rowflag[ry]) { // an all-zero staged row dilates to zeros; a row without kw non-zero bytes in a row erodes to zeros
            aE = 0; aO = 0;
        } else if constexpr (KWC > 0) {
            uint32_t r4 = dcw_hmax<KWC, OP == 1, MORPH_HALO>(tinw + ry * (IWB / 4), j);
            aE = r4 & 0x00FF00FFu; aO = (r4 >> 8) & 0x00FF00FFu;
        } else {
            int o = MORPH_HALO - ax + 4 * j; // byte offset of the window's first column
            const uint32_t *rw = tinw + ry * (IWB / 4);
            int qi = o >> 2;
            uint32_t lo = rw[qi], hi = rw[qi + 1];
            for (int dx = 0; dx < kw; dx++, o++) {
                if ((o >> 2) != qi) { qi = o >> 2; lo = hi; hi = rw[qi + 1]; }
                uint32_t sft = __builtin_amdgcn_alignbyte(hi, lo, (unsigned)(o & 3));
                us2 e = __builtin_bit_cast(us2, sft & 0x00FF00FFu), od = __builtin_bit_cast(us2, (sft >> 8) & 0x00FF00FFu);
                us2 ce = __builtin_bit_cast(us2, aE), co = __builtin_bit_cast(us2, aO);
                ce = OP ? __builtin_elementwise_min(ce, e) : __builtin_elementwise_max(ce, e);
                co = OP ? __builtin_elementwise_min(co, od) : __builtin_elementwise_max(co, od);
                aE = __builtin_bit_cast(uint32_t, ce);
                aO = __builtin_bit_cast(uint32_t, co);
            }
        }
        tmpE[it] = aE;
        tmpO[it] = aO;
    }
    __syncthreads();
    for (int it = threadIdx.x; it < MORPH_TH * (MORPH_TW / 4); it += 256) {
        int oy = it >> 4, j = it & 15;
        uint32_t aE = fsplit, aO = fsplit;
        const int khc = KHC > 0 ? KHC : kh;
        bool live = OP != 0; // dilation: some row of the window is live; erosion: every row of the window is
        if (!OP) for (int dy = 0; dy < khc; dy++) live = live || rowflag[oy + dy];
        else for (int dy = 0; dy < khc; dy++) live = live && rowflag[oy + dy];
        if (!live) { aE = 0; aO = 0; }
        else
#pragma unroll
            for (int dy = 0; dy < khc; dy++) {
                us2 e = __builtin_bit_cast(us2, tmpE[(oy + dy) * 16 + j]), od = __builtin_bit_cast(us2, tmpO[(oy + dy) * 16 + j]);
                us2 ce = __builtin_bit_cast(us2, aE), co = __builtin_bit_cast(us2, aO);
                ce = OP ? __builtin_elementwise_min(ce, e) : __builtin_elementwise_max(ce, e);
                co = OP ? __builtin_elementwise_min(co, od) : __builtin_elementwise_max(co, od);
                aE = __builtin_bit_cast(uint32_t, ce);
                aO = __builtin_bit_cast(uint32_t, co);
            }
        uint32_t word = aE | (aO << 8);
        if (lut)
            word = (uint32_t)slut[word & 0xff] | ((uint32_t)slut[(word >> 8) & 0xff] << 8) |
                   ((uint32_t)slut[(word >> 16) & 0xff] << 16) | ((uint32_t)slut[word >> 24] << 24);
        ((uint32_t *)tout)[it] = word;
    }
    __syncthreads();
    if (bits)
        for (int oy = wv; oy < MORPH_TH; oy += 4) {
            int gy = y0 + oy, gx = x0 + lane;
            u64 bal = __ballot(gy < h && gx < w && tout[oy * MORPH_TW + lane] != 0);
            if (lane == 0 && gy < h) bits[(size_t)g * h * wq + (size_t)gy * wq + bx] = bal;
        }
    if (threadIdx.x < MORPH_TH * 4) {
        int row = threadIdx.x >> 2, c16 = threadIdx.x & 3;
        int gy = y0 + row, gx = x0 + 16 * c16;
        if (gy < h && gx < w) {
            uint4 ov = *(const uint4 *)(tout + row * MORPH_TW + 16 * c16);
            *(uint4 *)(d + (size_t)gy * w + gx) = ov;
            // cell occupancy of the output for the consumer's tile list (k_dc_tiles): a lane's 16 bytes are one cell row
            if (cellout && (ov.x | ov.y | ov.z | ov.w)) {
                int cx = gx / CELLBM_COLS;
                atomicOr((unsigned long long *)&cellout[((size_t)g * bm_bands + gy / CELLBM_ROWS) * CELLBM_WORDS + (cx >> 6)], 1ull << (cx & 63));
            }
        }
    }
}

template <int OP, int KHC = 0, int KWC = 0>
__global__ void __launch_bounds__(256)
k_morph_rect_v(const uint8_t *src, uint8_t *dst, u64 *bits, const uint8_t *lut, int h, int w, int kh,
               int kw, const int *active, u64 *cellout, int bm_bands) {
    int g = blockIdx.z;
    if (active && !active[g]) return;
    morph_rect_tile<OP, KHC, KWC>(src, dst, bits, lut, h, w, kh, kw, cellout, bm_bands, blockIdx.x, blockIdx.y, g);
}

// The candidate tiles of a wide erosion of a sparse image (k_erode_cand below has zero-filled dst and left one bit per tile
// that can hold anything): one workgroup per tile ROW walks the set bits of the row's two mask words.  (A workgroup per tile
// that exits when its bit is clear spent the launch on 2 M workgroups per 256 LSST-size frames, 85 % of them empty.)
template <int OP, int KHC = 0, int KWC = 0>
__global__ void __launch_bounds__(256)
k_morph_rect_rows(const uint8_t *src, uint8_t *dst, const uint8_t *lut, int h, int w, int kh, int kw, const int *active,
                  const u64 *candmask, u64 *cellout, int bm_bands) {
    const int g = blockIdx.y, by = blockIdx.x;
    if (active && !active[g]) return;
    for (int half = 0; half < 2; half++) {
        u64 m = candmask[((size_t)g * gridDim.x + by) * 2 + half];
        while (m) {
            const int bx = 64 * half + __ffsll((long long)m) - 1;
            m &= m - 1;
            __syncthreads(); // the previous tile's LDS reads are done
            morph_rect_tile<OP, KHC, KWC>(src, dst, nullptr, lut, h, w, kh, kw, cellout, bm_bands, bx, by, g);
        }
    }
}

// Wide erosion (kw >= 7) of a sparse 8-bit image, first step.  The producer left one bit per aligned 4-pixel word ("all four
// bytes non-zero", k_prep_hist).  A kw-wide run of non-zero bytes contains such a word, so a 64 x 32 tile can only hold a
// non-zero output if kh consecutive rows of its input rows each have such a word within reach of the tile (words -1 .. 17
// of the tile's 16).  One wave per tile row: lane = tile column walks the rows counting consecutive live ones -> one
// candidate bit per tile (candmask, two u64 per tile row), and the wave zero-fills the tile row of dst with 16-byte stores.
// k_morph_rect_v then only runs on candidate tiles (a dim-pass sky: a handful around stars) instead of staging every tile.
__global__ void __launch_bounds__(64)
k_erode_cand(const u64 *fullbits, u64 *candmask, uint8_t *dst, int h, int w, int kh, const int *active, int fill) {
    const int g = blockIdx.y, ty = blockIdx.x, lane = threadIdx.x;
    if (active && !active[g]) return;
    const int y0 = ty * MORPH_TH, ay = kh / 2, IH = MORPH_TH + kh - 1, ntx = (w + MORPH_TW - 1) / MORPH_TW, fw = (w + 255) >> 8;
    for (int half = 0; half < 2; half++) {
        const int tx = lane + 64 * half;
        bool cand = false;
        if (tx < ntx) {
            const int j0 = 16 * tx - 1, jb = j0 < 0 ? 0 : j0, q = jb >> 6, sh2 = jb & 63, nb = 19 - (jb - j0);
            int consec = 0;
            const bool two = sh2 > 64 - 19 && q + 1 < fw; // the 19 bits straddle two words
            for (int t0 = 0; t0 < IH; t0 += 8) { // eight rows of loads in flight (a row at a time was a chain of 40 round trips)
                u64 b0[8], b1[8];
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const int gy = y0 - ay + t0 + k;
                    const bool in = t0 + k < IH && gy >= 0 && gy < h;
                    const u64 *fr = fullbits + ((size_t)g * h + (in ? gy : 0)) * fw;
                    b0[k] = in ? fr[q] : 0ull;
                    b1[k] = (in && two) ? fr[q + 1] : 0ull;
                }
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    if (t0 + k >= IH) break;
                    const int gy = y0 - ay + t0 + k;
                    bool live = gy < 0 || gy >= h; // rows outside the image are ignored by an erosion
                    if (!live) {
                        u64 b = b0[k] >> sh2;
                        if (two) b |= b1[k] << (64 - sh2);
                        live = (b & ((1ull << nb) - 1ull)) != 0ull;
                    }
                    consec = live ? consec + 1 : 0;
                    cand = cand || consec >= kh;
                }
            }
        }
        u64 m = __ballot(cand);
        if (lane == 0) candmask[((size_t)g * gridDim.x + ty) * 2 + half] = m;
    }
    if (!fill) return; // (k_fill_around zero-fills what the consumer can reach, once the erosion's output cells are known)
    uint4 *row = (uint4 *)(dst + (size_t)g * h * w);
    const int w16 = w >> 4;
    for (int r = y0; r < y0 + MORPH_TH && r < h; r++)
        for (int x = lane; x < w16; x += 64) row[(size_t)r * w16 + x] = make_uint4(0, 0, 0, 0);
}

// After the candidate tiles of a wide erosion have been written (k_morph_rect_rows, which also marked the 16 x 16 cells that
// hold anything): the only consumer, the fused dilate + Canny tile kernel, visits a 64 x 16 tile only if a marked cell lies
// within one cell band above / below and within cells 4 tx - 1 .. 4 tx + 4, and then reads rows 16 ty - 6 .. 16 ty + 21 (kh <= 9)
// and columns 64 tx - 16 .. 64 tx + 79.  So a non-candidate 64 x 32 erosion tile (TX, TY) can only be read if a marked cell
// lies in bands 2 TY - 2 .. 2 TY + 3 and cells 4 TX - 5 .. 4 TX + 8 (one more each way here): those are zero-filled, the rest
// of the plane -- most of a sky frame after a 9 x 9 erosion -- is left untouched instead of 16 MB of zeros per frame.
__global__ void __launch_bounds__(64)
k_fill_around(const u64 *candmask, const u64 *cellbm, int bm_bands, uint8_t *dst, int h, int w, const int *active) {
    const int g = blockIdx.y, ty = blockIdx.x, lane = threadIdx.x;
    if (active && !active[g]) return;
    const int ntx = (w + MORPH_TW - 1) / MORPH_TW;
    __shared__ u64 m[CELLBM_WORDS];
    if (lane < CELLBM_WORDS) {
        u64 v = 0ull;
        for (int b = 2 * ty - 3; b <= 2 * ty + 4; b++)
            if (b >= 0 && b < bm_bands) v |= cellbm[((size_t)g * bm_bands + b) * CELLBM_WORDS + lane];
        m[lane] = v;
    }
    __builtin_amdgcn_wave_barrier();
    uint8_t *d = dst + (size_t)g * h * w;
    for (int half = 0; half < 2; half++) {
        const int tx = lane + 64 * half;
        bool need = false;
        if (tx < ntx && !((candmask[((size_t)g * gridDim.x + ty) * 2 + half] >> lane) & 1ull)) {
            const int c0 = max(0, 4 * tx - 6), c1 = min(CELLBM_WORDS * 64 - 1, 4 * tx + 9); // cells c0 .. c1 (at most 16)
            u64 bits = m[c0 >> 6] >> (c0 & 63);
            if ((c0 & 63) + (c1 - c0) > 63 && (c0 >> 6) + 1 < CELLBM_WORDS) bits |= m[(c0 >> 6) + 1] << (64 - (c0 & 63));
            need = (bits & ((2ull << (c1 - c0)) - 1ull)) != 0ull;
        }
        for (u64 todo = __ballot(need); todo; todo &= todo - 1) {
            const int t = __ffsll((long long)todo) - 1 + 64 * half, x0 = t * MORPH_TW;
            for (int it = lane; it < MORPH_TH * (MORPH_TW / 16); it += 64) { // 32 rows x 4 pieces of 16 bytes
                const int r = ty * MORPH_TH + (it >> 2), x = x0 + 16 * (it & 3);
                if (r < h && x < w) *(uint4 *)(d + (size_t)r * w + x) = make_uint4(0, 0, 0, 0);
            }
        }
    }
}

// arbitrary 0/1 structuring element (the knob is an array: detecttrails.py:205,220-221)
template <int OP>
__global__ void __launch_bounds__(256)
k_morph_generic(const uint8_t *src, uint8_t *dst, u64 *bits, const uint8_t *lut,
                const uint8_t *kernel, int h, int w, int kh, int kw, const int *active) {
    int g = blockIdx.z;
    if (active && !active[g]) return;
    int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int gx = blockIdx.x * 64 + lane, gy = blockIdx.y * 4 + wv;
    size_t N = (size_t)h * w;
    const uint8_t *s = src + (size_t)g * N;
    int ay = kh / 2, ax = kw / 2;
    int m = OP ? 255 : 0;
    bool valid = gx < w && gy < h;
    if (valid)
        for (int dy = 0; dy < kh; dy++) {
            int yy = gy + dy - ay;
            if (yy < 0 || yy >= h) continue;
            for (int dx = 0; dx < kw; dx++) {
                if (!kernel[dy * kw + dx]) continue;
                int xx = gx + dx - ax;
                if (xx < 0 || xx >= w) continue;
                int v = s[(size_t)yy * w + xx];
                m = OP ? min(m, v) : max(m, v);
            }
        }
    int out = lut ? lut[g * 256 + m] : m;
    if (valid) dst[(size_t)g * N + (size_t)gy * w + gx] = (uint8_t)out;
    if (bits) {
        int wq = LFD_WQ(w);
        u64 bal = __ballot(valid && out != 0);
        if (lane == 0 && gy < h) bits[(size_t)g * h * wq + (size_t)gy * wq + blockIdx.x] = bal;
    }
}

// ------------------------------------------------------------------------------------------
// Optional Gaussian smoothing of Canny's input (off by default: cv2.Canny has no such stage; see include/lfdmi.h).
// Separable float32 filter, taps from the host (cv2.getGaussianKernel's formula), BORDER_REFLECT_101, accumulation
// in tap order without FMA, one rounding to 8 bit at the end.  64 x 16 tile per workgroup: the u8 tile with its
// halo and the horizontal float result live in LDS.
// ------------------------------------------------------------------------------------------
#define GAUSS_TW 64
#define GAUSS_TH 16
struct GaussTaps { float k[32]; int n; };

__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}

__global__ void __launch_bounds__(256)
k_gauss(const uint8_t *src, uint8_t *dst, int h, int w, GaussTaps taps, const int *active) {
    int g = blockIdx.z;
    if (active && !active[g]) return;
    extern __shared__ __attribute__((aligned(16))) uint8_t smg[];
    const int n = taps.n, r = n / 2;
    const int IW = GAUSS_TW + 2 * r, IH = GAUSS_TH + 2 * r;
    uint8_t *tin = smg;                                              // IH x IW bytes
    float *hor = (float *)(smg + ((IH * IW + 15) & ~15));            // IH x GAUSS_TW floats
    const int x0 = blockIdx.x * GAUSS_TW, y0 = blockIdx.y * GAUSS_TH;
    const uint8_t *s = src + (size_t)g * h * w;
    for (int idx = threadIdx.x; idx < IH * IW; idx += 256) {
        int iy = idx / IW, ix = idx - iy * IW;
        tin[idx] = s[(size_t)reflect101(y0 + iy - r, h) * w + reflect101(x0 + ix - r, w)];
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < IH * GAUSS_TW; idx += 256) {
        int iy = idx / GAUSS_TW, ix = idx - iy * GAUSS_TW;
        float acc = 0.f;
        for (int k = 0; k < n; k++) acc = __fadd_rn(acc, __fmul_rn((float)tin[iy * IW + ix + k], taps.k[k]));
        hor[idx] = acc;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < GAUSS_TH * GAUSS_TW; idx += 256) {
        int oy = idx / GAUSS_TW, ox = idx - oy * GAUSS_TW;
        int gy = y0 + oy, gx = x0 + ox;
        if (gy >= h || gx >= w) continue;
        float acc = 0.f;
        for (int k = 0; k < n; k++) acc = __fadd_rn(acc, __fmul_rn(hor[(oy + k) * GAUSS_TW + ox], taps.k[k]));
        dst[(size_t)g * h * w + (size_t)gy * w + gx] = (uint8_t)sat_u8_f32(acc);
    }
}

// u8 image -> "!= 0" bit rows (standalone HoughLines entry point)
__global__ void __launch_bounds__(256)
k_bits_from_u8(const uint8_t *src, u64 *bits, int h, int w) {
    int g = blockIdx.z;
    int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int gx = blockIdx.x * 64 + lane, gy = blockIdx.y * 4 + wv;
    bool valid = gx < w && gy < h;
    int v = valid ? src[(size_t)g * h * w + (size_t)gy * w + gx] : 0;
    u64 bal = __ballot(v != 0);
    int wq = LFD_WQ(w);
    if (lane == 0 && gy < h) bits[(size_t)g * h * wq + (size_t)gy * wq + blockIdx.x] = bal;
}

// bit rows -> 0/255 u8 image
__global__ void __launch_bounds__(256)
k_u8_from_bits(const u64 *bits, uint8_t *dst, int h, int w) {
    int g = blockIdx.z;
    int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int gx = blockIdx.x * 64 + lane, gy = blockIdx.y * 4 + wv;
    int wq = LFD_WQ(w);
    if (gx < w && gy < h) {
        u64 b = bits[(size_t)g * h * wq + (size_t)gy * wq + blockIdx.x];
        dst[(size_t)g * h * w + (size_t)gy * w + gx] = ((b >> lane) & 1ull) ? 255 : 0;
    }
}

// ------------------------------------------------------------------------------------------
// Canny, first half: Sobel 3x3 (replicate border) -> L1 magnitude -> non-maximum suppression
// with OpenCV's fixed-point tan(22.5 deg) sector test -> two bit rows per image row:
// cand (passes NMS and mag > low) and strong (cand and mag > high).  64 x 16 tile per
// workgroup staged in LDS with a 2-px halo; one wave == 64 consecutive columns so a
// __ballot is exactly one output word.
// ------------------------------------------------------------------------------------------
#define CANNY_TW 64
#define CANNY_TH 32
#define CANNY_HALO 16               // staged columns left/right of the tile (aligned 16-byte pieces)
#define CANNY_MOFF 12               // first staged column that gets a magnitude (tile column -4)
#define CANNY_MW 72                 // magnitudes per row: tile columns -4 .. 67

// Canny stages on a staged tile: px = 36 rows x 96 bytes (rows y0-2.., cols x0-16..), rowflag = row
// holds a non-zero byte.  Sobel + L1 magnitude four pixels per lane, then NMS -> bit rows.
__device__ __forceinline__ void canny_tile_stages(const uint8_t *px, const int *rowflag, int *mg, int *dxy, u64 *cand,
                                                  u64 *strong, int g, int h, int w, int x0, int y0, int low, int high) {
    const int PWB = CANNY_TW + 2 * CANNY_HALO, MH = CANNY_TH + 2;
    const int wq = LFD_WQ(w);
    // Sobel + L1 magnitude, 4 pixels per lane: staged bytes 12..83 of rows 1..MH (tile cols -4..67)
    const uint32_t *pw = (const uint32_t *)px;
    for (int it = threadIdx.x; it < MH * (CANNY_MW / 4); it += 256) {
        int my = it / (CANNY_MW / 4), k = it - my * (CANNY_MW / 4); // k: word 3 + k of the staged row
        int gy = y0 - 1 + my;
        int4 m4 = make_int4(0, 0, 0, 0), d4 = make_int4(0, 0, 0, 0);
        if ((rowflag[my] | rowflag[my + 1] | rowflag[my + 2]) && gy >= 0 && gy < h) {
            int b[3][6]; // bytes -1..4 around the word, for the three rows
#pragma unroll
            for (int r = 0; r < 3; r++) {
                const uint32_t *rw = pw + (my + r) * (PWB / 4) + 3 + k;
                uint32_t wm = rw[-1], wc = rw[0], wp = rw[1];
                b[r][0] = wm >> 24; b[r][1] = wc & 0xff; b[r][2] = (wc >> 8) & 0xff;
                b[r][3] = (wc >> 16) & 0xff; b[r][4] = wc >> 24; b[r][5] = wp & 0xff;
            }
            int mm[4], dd[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                int gx = x0 - 4 + 4 * k + i;
                int dx = (b[0][i + 2] - b[0][i]) + 2 * (b[1][i + 2] - b[1][i]) + (b[2][i + 2] - b[2][i]);
                int dy = (b[2][i] - b[0][i]) + 2 * (b[2][i + 1] - b[0][i + 1]) + (b[2][i + 2] - b[0][i + 2]);
                bool in = gx >= 0 && gx < w;
                mm[i] = in ? abs(dx) + abs(dy) : 0;
                dd[i] = in ? ((dx & 0xffff) | (dy << 16)) : 0;
            }
            m4 = make_int4(mm[0], mm[1], mm[2], mm[3]);
            d4 = make_int4(dd[0], dd[1], dd[2], dd[3]);
        }
        ((int4 *)mg)[it] = m4;
        ((int4 *)dxy)[it] = d4;
    }
    __syncthreads();
    int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int oy = wv; oy < CANNY_TH; oy += 4) {
        int gy = y0 + oy, gx = x0 + lane;
        bool keep = false, str = false;
        if (gy < h && gx < w) {
            const int *mc = mg + (oy + 1) * CANNY_MW + (lane + 4);
            int m = mc[0];
            if (m > low) {
                int d = dxy[(oy + 1) * CANNY_MW + (lane + 4)];
                int xs = (int)(short)(d & 0xffff), ys = d >> 16;
                int ax = abs(xs), ay = abs(ys) << 15;
                int tg22x = ax * 13573;
                if (ay < tg22x) {
                    keep = (m > mc[-1]) && (m >= mc[1]);
                } else {
                    int tg67x = tg22x + (ax << 16);
                    if (ay > tg67x) {
                        keep = (m > mc[-CANNY_MW]) && (m >= mc[CANNY_MW]);
                    } else {
                        int sgn = ((xs ^ ys) < 0) ? -1 : 1;
                        keep = (m > mc[-CANNY_MW - sgn]) && (m > mc[CANNY_MW + sgn]);
                    }
                }
                str = keep && (m > high);
            }
        }
        u64 bc = __ballot(keep), bs = __ballot(str);
        if (lane == 0 && gy < h) {
            size_t o = (size_t)g * h * wq + (size_t)gy * wq + blockIdx.x;
            cand[o] = bc;
            strong[o] = bs;
        }
    }
}

// Frames whose width is a multiple of 16: 16-byte tile loads, all-zero-tile exit before any LDS
// traffic, Sobel/magnitude computed for four pixels per lane from dword LDS reads.
__global__ void __launch_bounds__(256)
k_canny_nms_v(const uint8_t *img, u64 *cand, u64 *strong, int h, int w, int low, int high,
              const int *active) {
    int g = blockIdx.z;
    if (active && !active[g]) return;
    const int PH = CANNY_TH + 4, PWB = CANNY_TW + 2 * CANNY_HALO, MH = CANNY_TH + 2;
    __shared__ __attribute__((aligned(16))) uint8_t px[PH * PWB]; // rows y0-2 .. y0+TH+1, cols x0-16 .. x0+79
    __shared__ __attribute__((aligned(16))) int mg[MH * CANNY_MW];
    __shared__ __attribute__((aligned(16))) int dxy[MH * CANNY_MW];
    __shared__ int rowflag[PH];
    int x0 = blockIdx.x * CANNY_TW, y0 = blockIdx.y * CANNY_TH;
    const uint8_t *s = img + (size_t)g * h * w;
    if (threadIdx.x < PH) rowflag[threadIdx.x] = 0;
    __syncthreads();
    // one 16-byte piece per thread (36 rows x 6 pieces = 216), BORDER_REPLICATE by clamping
    uint4 v = make_uint4(0, 0, 0, 0);
    int ty = threadIdx.x / 6, wx = threadIdx.x - ty * 6;
    bool mine = threadIdx.x < PH * 6;
    if (mine) {
        int gy = min(max(y0 - 2 + ty, 0), h - 1), gx = x0 - CANNY_HALO + 16 * wx;
        if (gx >= 0 && gx < w) v = *(const uint4 *)(s + (size_t)gy * w + gx);
        else {
            uint32_t e = s[(size_t)gy * w + (gx < 0 ? 0 : w - 1)];
            e |= e << 8; e |= e << 16;
            v = make_uint4(e, e, e, e);
        }
    }
    uint32_t nzv = v.x | v.y | v.z | v.w;
    int wq = LFD_WQ(w);
    // an all-zero tile (sky) has zero gradient everywhere: no candidate when low >= 0
    if (!__syncthreads_or(nzv != 0) && low >= 0) {
        if (threadIdx.x < CANNY_TH && y0 + threadIdx.x < h) {
            size_t o = (size_t)g * h * wq + (size_t)(y0 + threadIdx.x) * wq + blockIdx.x;
            cand[o] = 0ull;
            strong[o] = 0ull;
        }
        return;
    }
    if (mine) {
        ((uint4 *)px)[threadIdx.x] = v;
        if (nzv) rowflag[ty] = 1;
    }
    __syncthreads();
    canny_tile_stages(px, rowflag, mg, dxy, cand, strong, g, h, w, x0, y0, low, high);
}

// ------------------------------------------------------------------------------------------
// Fused dilation + Canny NMS (batch path): the dilated, equalised tile with a 2-pixel ring is
// built in LDS and handed straight to the Sobel / NMS stages, so `equ` is never re-read (and
// not even written unless stage images are requested).  Same arithmetic as k_morph_rect_v<0>
// followed by k_canny_nms_v, bit for bit: the ring positions outside the image take the
// replicated border value of the dilated image (Canny's BORDER_REPLICATE), positions inside
// are ordinary dilation outputs computed from a correspondingly larger input halo.
// Requires w % 16 == 0 and kw/2 + 2 <= 16, kw - 1 - kw/2 + 2 <= 16.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_dilate_canny_v(const uint8_t *src, uint8_t *equ, u64 *equb, u64 *cand, u64 *strong, const uint8_t *lut, int h, int w,
                 int kh, int kw, int low, int high, const int *active) {
    int g = blockIdx.z;
    if (active && !active[g]) return;
    extern __shared__ __attribute__((aligned(16))) uint8_t smf[];
    const int PH = CANNY_TH + 4, PWB = CANNY_TW + 2 * CANNY_HALO, MH = CANNY_TH + 2;
    const int IH = PH + kh - 1;          // input rows feeding dilated rows -2 .. TH+1
    const int NWD = CANNY_MW / 4;        // 18 dwords per dilated row: tile columns -4 .. 67
    __shared__ __attribute__((aligned(16))) uint8_t px[PH * PWB];
    __shared__ uint8_t slut[256];
    __shared__ int rowflag[CANNY_TH + 4 + LFDMI_MAX_MORPH_K]; // input rows
    __shared__ int pxflag[PH];                                 // dilated rows
    uint8_t *tin = smf;                                        // IH x PWB
    uint32_t *tmpE = (uint32_t *)(smf + IH * PWB), *tmpO = tmpE + IH * NWD;
    int x0 = blockIdx.x * CANNY_TW, y0 = blockIdx.y * CANNY_TH;
    int ay = kh / 2, ax = kw / 2;
    size_t N = (size_t)h * w;
    const uint8_t *s = src + (size_t)g * N;
    if (lut) slut[threadIdx.x] = lut[g * 256 + threadIdx.x];
    if (threadIdx.x < IH) rowflag[threadIdx.x] = 0;
    if (threadIdx.x < PH) pxflag[threadIdx.x] = 0;
    __syncthreads();
    uint32_t any = 0;
    for (int idx = threadIdx.x; idx < IH * 6; idx += 256) {
        int iy = idx / 6, wx = idx - iy * 6;
        int gy = y0 - 2 - ay + iy, gx = x0 - CANNY_HALO + 16 * wx;
        uint4 v = make_uint4(0, 0, 0, 0); // out-of-image samples are ignored by a dilation
        if (gy >= 0 && gy < h && gx >= 0 && gx < w) v = *(const uint4 *)(s + (size_t)gy * w + gx);
        uint32_t nzv = v.x | v.y | v.z | v.w;
        if (nzv) rowflag[iy] = 1;
        any |= nzv;
        ((uint4 *)tin)[idx] = v;
    }
    int wq = LFD_WQ(w);
    int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int nz = __syncthreads_or(any != 0);
    uint8_t *d = equ ? equ + (size_t)g * N : nullptr;
    if (!nz) {
        // empty neighbourhood: the dilated tile and its ring are lut[0] everywhere: no gradient
        uint32_t z = lut ? slut[0] : 0u;
        z |= z << 8; z |= z << 16;
        if (d && threadIdx.x < CANNY_TH * 4) {
            int row = threadIdx.x >> 2, c16 = threadIdx.x & 3;
            int gy = y0 + row, gx = x0 + 16 * c16;
            if (gy < h && gx < w) *(uint4 *)(d + (size_t)gy * w + gx) = make_uint4(z, z, z, z);
        }
        if (threadIdx.x < CANNY_TH && y0 + threadIdx.x < h) {
            size_t o = (size_t)g * h * wq + (size_t)(y0 + threadIdx.x) * wq + blockIdx.x;
            equb[o] = z ? valid_mask(blockIdx.x, w) : 0ull;
            cand[o] = 0ull;
            strong[o] = 0ull;
        }
        if (low >= 0) return;
        // low < 0 makes zero-gradient pixels candidates: fall through to the general path
    }
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    const uint32_t *tinw = (const uint32_t *)tin;
    for (int it = threadIdx.x; it < IH * NWD; it += 256) {
        int ry = it / NWD, j = it - ry * NWD;
        uint32_t aE = 0, aO = 0;
        if (rowflag[ry]) {
            int o = CANNY_MOFF - ax + 4 * j; // byte offset of the window's first column
            const uint32_t *rw = tinw + ry * (PWB / 4);
            int qi = o >> 2;
            uint32_t lo = rw[qi], hi = rw[qi + 1];
            for (int dx = 0; dx < kw; dx++, o++) {
                if ((o >> 2) != qi) { qi = o >> 2; lo = hi; hi = rw[qi + 1]; }
                uint32_t sft = __builtin_amdgcn_alignbyte(hi, lo, (unsigned)(o & 3));
                us2 e = __builtin_bit_cast(us2, sft & 0x00FF00FFu), od = __builtin_bit_cast(us2, (sft >> 8) & 0x00FF00FFu);
                aE = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(us2, aE), e));
                aO = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(us2, aO), od));
            }
        }
        tmpE[it] = aE;
        tmpO[it] = aO;
    }
    __syncthreads();
    for (int it = threadIdx.x; it < PH * NWD; it += 256) {
        int dr = it / NWD, j = it - dr * NWD;
        uint32_t aE = 0, aO = 0;
        bool live = false;
        for (int dy = 0; dy < kh; dy++) live = live || rowflag[dr + dy];
        if (live)
            for (int dy = 0; dy < kh; dy++) {
                aE = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(us2, aE), __builtin_bit_cast(us2, tmpE[(dr + dy) * NWD + j])));
                aO = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(us2, aO), __builtin_bit_cast(us2, tmpO[(dr + dy) * NWD + j])));
            }
        uint32_t word = aE | (aO << 8);
        if (lut)
            word = (uint32_t)slut[word & 0xff] | ((uint32_t)slut[(word >> 8) & 0xff] << 8) |
                   ((uint32_t)slut[(word >> 16) & 0xff] << 16) | ((uint32_t)slut[word >> 24] << 24);
        ((uint32_t *)px)[dr * (PWB / 4) + 3 + j] = word;
        if (word) pxflag[dr] = 1;
    }
    __syncthreads();
    // ring positions outside the image: replicate the dilated image's border (tiles on the frame only)
    if (y0 < 2 || y0 + CANNY_TH + 2 > h || x0 < 4 || x0 + CANNY_TW + 4 > w) {
        for (int it = threadIdx.x; it < PH * CANNY_MW; it += 256) {
            int dr = it / CANNY_MW, c = CANNY_MOFF + it - dr * CANNY_MW;
            int gy = y0 - 2 + dr, gx = x0 - CANNY_HALO + c;
            if (gy < 0 || gy >= h || gx < 0 || gx >= w) {
                int cy = min(max(gy, 0), h - 1), cx = min(max(gx, 0), w - 1);
                uint8_t v = px[(cy - (y0 - 2)) * PWB + (cx - (x0 - CANNY_HALO))];
                px[dr * PWB + c] = v;
                if (v) pxflag[dr] = 1;
            }
        }
        __syncthreads();
    }
    // the tile proper: equ (optional) and its != 0 bit row
    if (d && threadIdx.x < CANNY_TH * 4) {
        int row = threadIdx.x >> 2, c16 = threadIdx.x & 3;
        int gy = y0 + row, gx = x0 + 16 * c16;
        if (gy < h && gx < w) *(uint4 *)(d + (size_t)gy * w + gx) = *(const uint4 *)(px + (row + 2) * PWB + CANNY_HALO + 16 * c16);
    }
    for (int oy = wv; oy < CANNY_TH; oy += 4) {
        int gy = y0 + oy, gx = x0 + lane;
        u64 bal = __ballot(gy < h && gx < w && px[(oy + 2) * PWB + CANNY_HALO + lane] != 0);
        if (lane == 0 && gy < h) equb[(size_t)g * h * wq + (size_t)gy * wq + blockIdx.x] = bal;
    }
    __syncthreads(); // tin / tmp are dead: reuse the dynamic LDS for the Canny stages
    int *mg = (int *)smf, *dxy = mg + MH * CANNY_MW;
    canny_tile_stages(px, pxflag, mg, dxy, cand, strong, g, h, w, x0, y0, low, high);
}

// ------------------------------------------------------------------------------------------
// Fused dilation + Canny NMS, one WAVE per 64 x 16 tile, driven by activity masks.
//
// The pass input is sparse (well under 1 % non-zero bytes, yet a third of the tiles hold
// some), so sweeping every tile position wastes the vector units, and a four-wave workgroup
// that meets at a barrier after every stage mostly waits.  Here a workgroup IS one wave with
// its own LDS tile: no cross-wave barrier, many independent waves per CU to hide latency.
// Per tile the wave keeps one 32-bit mask per 4-pixel word column (bit r = "row r of this
// column can differ from the background"), pushes the masks through the stages with shifts
// and ORs (input pieces -> horizontal max -> vertical max -> Sobel -> NMS) and evaluates only
// live (row, column) items, two columns per step (lanes 0-31 / 32-63 = rows of each).
// Items a later stage reads but that were not evaluated hold the background value by
// construction: px is pre-filled with lut[0], the magnitude plane with 0.
// Same arithmetic as k_morph_rect_v<0> followed by k_canny_nms_v, bit for bit.
// Requires w % 16 == 0, kw/2 + 4 <= 16, kw - 1 - kw/2 + 4 <= 16, kh <= 13, low >= 0.
// The grid is 1-D and XCD-aware: frame = 8 * (j / tiles_per_frame) + (block & 7) keeps every
// tile of a frame (and its halo re-reads) behind one L2; a wave walks a strip of S tiles
// downwards and prefetches the next tile's input while it works on the current one.
// ------------------------------------------------------------------------------------------
#define DCW_TH 16
#define DCW_PH (DCW_TH + 4)  // dilated rows -2 .. TH+1
#define DCW_MH (DCW_TH + 2)  // magnitude rows -1 .. TH
#define DCW_TS 112           // tin / px row stride in bytes: 96 used, 16-byte aligned, rows spread over the banks
#define DCW_NWD 18           // word columns per row: tile columns -4 .. 67
#define DCW_MAXKH 13         // DCW_PH + kh - 1 <= 32 rows: one 32-bit mask per column
#define DCW_MAXS 8            // tiles per strip (one wave walks a strip left to right)
static_assert(DCW_TH == CELLBM_ROWS && CANNY_TW == 4 * CELLBM_COLS, "a tile is one band high and four cells wide");

template <class F>
__device__ __forceinline__ void dcw_for_live(u64 live, uint32_t colmask, int lane, F f) {
    int r = lane & 31;
    bool up = lane >= 32;
    while (live) {
        int ja = __ffsll((long long)live) - 1;
        live &= live - 1;
        uint32_t ma = __builtin_amdgcn_readlane(colmask, ja), mb = 0;
        int jb = ja;
        if (live) {
            jb = __ffsll((long long)live) - 1;
            live &= live - 1;
            mb = __builtin_amdgcn_readlane(colmask, jb);
        }
        uint32_t m = up ? mb : ma;
        if ((m >> r) & 1) f(r, up ? jb : ja);
    }
}

// The same with ROWS lanes per column instead of 32: 64 / ROWS live columns per pass.  The Sobel items of a tile have 18 rows
// and the NMS items 16 (rows R0 .. R0 + ROWS - 1): three / four columns per pass instead of two with half the lanes idle.
template <int ROWS, int R0, typename F>
__device__ __forceinline__ void dcw_for_live_n(u64 live, uint32_t colmask, int lane, F f) {
    constexpr int NG = 64 / ROWS;
    const int gi = lane / ROWS, r = R0 + lane - gi * ROWS;
    while (live) {
        uint32_t m = 0;
        int j = 0;
#pragma unroll
        for (int t = 0; t < NG; t++)
            if (live) {
                const int jt = __ffsll((long long)live) - 1;
                live &= live - 1;
                const uint32_t mt = __builtin_amdgcn_readlane(colmask, jt);
                if (gi == t) { m = mt; j = jt; }
            }
        if (gi < NG && ((m >> r) & 1)) f(r, j);
    }
}

__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(5, 8)))
k_dilate_canny_w(const uint8_t *src, uint8_t *equ, u64 *equb, u64 *cand, u64 *strong, const uint8_t *lut, int h, int w,
                 int kh, int kw, int low, int high, const int *active, int nc, int tiles_x, int nstripx, int S, int SS,
                 const u64 *cellbm, int bm_bands) {
    const int tiles_y = (h + DCW_TH - 1) / DCW_TH;
    int L = blockIdx.x, xcd = L & 7, jb_ = L >> 3, per = tiles_y * nstripx;
    int g = (jb_ / per) * 8 + xcd;
    if (g >= nc) return;
    if (active && !active[g]) return;
    int rem = jb_ - (jb_ / per) * per;
    const int ty = rem / nstripx, txs = (rem - ty * nstripx) * S * SS; // SS strips of S tiles, left to right
    extern __shared__ __attribute__((aligned(16))) uint8_t smw[];
    const int PH = DCW_PH, MH = DCW_MH, NWD = DCW_NWD, TS = DCW_TS, TSW = DCW_TS / 4;
    const int IH = PH + kh - 1;
    const int MGB = MH * CANNY_MW * 2;                 // magnitude plane: ushort (m << 2 | sector)
    const int tin_bytes = IH * TS > MGB ? IH * TS : MGB;
    uint8_t *tin = smw;                                // IH x 112; later the magnitude plane
    uint32_t *tmp = (uint32_t *)(smw + tin_bytes);      // IH x 18 words of horizontal maxima
    uint8_t *px = (uint8_t *)(tmp + IH * NWD);         // PH x 112
    unsigned short *mg = (unsigned short *)smw;
    __shared__ uint8_t slut[256];
    __shared__ uint32_t Pm[8];                         // input piece column p: rows holding a non-zero byte
    __shared__ u64 rowc[DCW_TH], rows[DCW_TH];         // NMS output bit rows of the tile
    __shared__ u64 rowe[DCW_TH];                       // != 0 bits of the dilated, equalised tile
    const int lane = threadIdx.x;
    const int ay = kh / 2, ax = kw / 2;
    const size_t N = (size_t)h * w;
    const uint8_t *s = src + (size_t)g * N;
    const int wq = LFD_WQ(w);
    uint8_t *d = equ ? equ + (size_t)g * N : nullptr;
    const int npieces = IH * 6;                        // <= 192: three 16-byte pieces per lane
    const int y0 = ty * DCW_TH;
    int iy[3], wx[3];
    bool rowok[3], has[3];
#pragma unroll
    for (int p = 0; p < 3; p++) {
        int idx = lane + 64 * p;
        iy[p] = idx / 6;
        wx[p] = idx - iy[p] * 6;
        has[p] = idx < npieces;
        int gy = y0 - 2 - ay + iy[p];
        rowok[p] = has[p] && gy >= 0 && gy < h;
    }
#pragma unroll
    for (int p = 0; p < 4; p++) slut[lane + 64 * p] = lut ? lut[g * 256 + lane + 64 * p] : (uint8_t)(lane + 64 * p);
    if (lane < 8) Pm[lane] = 0u;
    __syncthreads();
    uint32_t zw = slut[0]; // background of the dilated, equalised image
    zw |= zw << 8; zw |= zw << 16;
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    const uint32_t mPH = (1u << PH) - 1, mMH = (1u << MH) - 1, mIH = IH >= 32 ? ~0u : ((1u << IH) - 1);
    for (int sub = 0; sub < SS; sub++) {
    const int tx0 = txs + sub * S;
    if (tx0 >= tiles_x) break;
    const int ntile = min(S, tiles_x - tx0); // tiles of this strip: tx0 .. tx0 + ntile - 1
    // Cell occupancy from the prep kernel (cell_mark): which 16-pixel x 16-row cells of the input hold a
    // non-zero byte.  A tile reads columns x0 - 16 .. x0 + 79 (cells 4 tx - 1 .. 4 tx + 4) of rows inside the
    // bands ty - 1 .. ty + 1; if those 18 cells are clear its outputs are written without loading anything.
    // bit j of `busy` = cell column 4 tx0 - 1 + j is occupied in one of the three bands.
    u64 busy = ~0ull;
    if (cellbm) {
        u64 f = 0;
        int band = ty - 1 + lane;
        if (lane < 3 && band >= 0 && band < bm_bands) {
            const u64 *bw = cellbm + ((size_t)g * bm_bands + band) * CELLBM_WORDS;
            int c0 = 4 * tx0 - 1, cs = max(c0, 0), wi = cs >> 6, sh = cs & 63;
            f = bw[wi] >> sh;
            if (sh && wi + 1 < CELLBM_WORDS) f |= bw[wi + 1] << (64 - sh);
            if (c0 < 0) f <<= 1;
        }
        unsigned lo = (unsigned)f, hi = (unsigned)(f >> 32);
        lo = __builtin_amdgcn_readlane(lo, 0) | __builtin_amdgcn_readlane(lo, 1) | __builtin_amdgcn_readlane(lo, 2);
        hi = __builtin_amdgcn_readlane(hi, 0) | __builtin_amdgcn_readlane(hi, 1) | __builtin_amdgcn_readlane(hi, 2);
        busy = ((u64)hi << 32) | lo;
    }
    auto needed = [&](int t) { return ((busy >> (4 * t)) & 0x3Full) != 0; };
    auto load_tile = [&](int t, uint4 *v) {
        int xl = (tx0 + t) * CANNY_TW - CANNY_HALO;
#pragma unroll
        for (int p = 0; p < 3; p++) {
            v[p] = make_uint4(0, 0, 0, 0); // out-of-image samples are ignored by a dilation
            int gx = xl + 16 * wx[p];
            if (rowok[p] && gx >= 0 && gx < w) v[p] = *(const uint4 *)(s + (size_t)(y0 - 2 - ay + iy[p]) * w + gx);
        }
    };
    uint4 v[3];
    v[0] = v[1] = v[2] = make_uint4(0, 0, 0, 0);
    if (needed(0)) load_tile(0, v);
    for (int t = 0; t < ntile; t++) {
        const int tx = tx0 + t, x0 = tx * CANNY_TW;
        uint32_t anyv = 0;
        if (needed(t)) {
#pragma unroll
            for (int p = 0; p < 3; p++)
                if (has[p]) {
                    uint32_t nzv = v[p].x | v[p].y | v[p].z | v[p].w;
                    if (nzv) atomicOr(&Pm[wx[p]], 1u << iy[p]);
                    anyv |= nzv;
                    *(uint4 *)(tin + iy[p] * TS + wx[p] * 16) = v[p];
                }
        }
        if (t + 1 < ntile && needed(t + 1)) load_tile(t + 1, v); // next tile's input: in flight while this one is processed
        if (__ballot(anyv != 0) == 0ull) {
            // empty neighbourhood: the dilated tile and its ring are lut[0] everywhere: no gradient
            if (d) {
                int row = lane >> 2, c16 = lane & 3;
                int gy = y0 + row, gx = x0 + 16 * c16;
                if (gy < h && gx < w) *(uint4 *)(d + (size_t)gy * w + gx) = make_uint4(zw, zw, zw, zw);
            }
            if (lane < DCW_TH && y0 + lane < h) {
                size_t o = (size_t)g * h * wq + (size_t)(y0 + lane) * wq + tx;
                equb[o] = zw ? valid_mask(tx, w) : 0ull;
                cand[o] = 0ull;
                strong[o] = 0ull;
            }
            continue;
        }
#define DCW_STAMP(k)
        constexpr int DCW_KHC = 0, DCW_KWC = 0; // (structuring element size only known at run time here)
#include "k_dcw_tile.inc"
#undef DCW_STAMP
    }
    }
}

// ------------------------------------------------------------------------------------------
// The same fused stages driven by a list of ACTIVE tiles.
//
// On sky frames 55-65 % of the 64 x 16 tiles have nothing within reach (their 6 x 3 cells of the occupancy bitmap are
// clear), and a wave that walks a strip spends most of its time on them: three scattered 8-byte stores per row of every
// empty tile, loop overhead, and waves whose strips are empty sit next to waves whose strips are full.  So:
//   k_dc_tiles        one workgroup per frame turns the cell bitmap into the frame's list of active tiles (raster
//                     order) and writes the BACKGROUND of the three output bit planes (equ != 0, candidates, strong)
//                     for the whole frame with coalesced 16-byte stores;
//   k_dilate_canny_t  `parts` waves per frame each take a contiguous share of the list and run the tile stages on
//                     it (input of the next tile in flight while the current one is processed), overwriting the
//                     words of their tiles.  Every wave has real work on every step.
// Results are identical to k_dilate_canny_w (same per-tile code, k_dcw_tile.inc).
// ------------------------------------------------------------------------------------------
#define DCT_THREADS 1024
#define DCT_MAXBANDS 512 // 8191 rows / 16

__device__ __forceinline__ unsigned dct_cells(const u64 *m, int b0) { // bits b0 .. b0 + 5 of a 512-bit row mask (b0 may be -1)
    int st = b0 < 0 ? 0 : b0, wi = st >> 6, sh = st & 63;
    u64 f = m[wi] >> sh;
    if (sh > 58 && wi + 1 < CELLBM_WORDS) f |= m[wi + 1] << (64 - sh);
    if (b0 < 0) f <<= 1;
    return (unsigned)(f & 0x3Full);
}

__global__ void __launch_bounds__(DCT_THREADS)
k_dc_tiles(const u64 *cellbm, int bm_bands, const uint8_t *lut, int *tile_list, int tile_cap, int *counters, u64 *equb, u64 *cand,
           u64 *strong, uint8_t *equ, int h, int w, const int *active, const int *perm) {
    // grid (frames, fill parts): part 0 builds the frame's tile list, every part writes its share of the background
    const int g = perm ? perm[blockIdx.x] : (int)blockIdx.x; // (perm: the pass's active frames first, see k_active_perm)
    if (active && !active[g]) return;
    const int fpart = blockIdx.y, fparts = gridDim.y;
    const int tiles_x = (w + CANNY_TW - 1) / CANNY_TW, tiles_y = (h + DCW_TH - 1) / DCW_TH;
    __shared__ int band_off[DCT_MAXBANDS + 1];
    __shared__ u64 band_mask[DCT_MAXBANDS][2];
    __shared__ u64 rowm[DCT_THREADS / 64][CELLBM_WORDS];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = DCT_THREADS / 64;
    if (fpart == 0) { // (uniform per workgroup: the barriers inside are safe)
        // ---- which tiles have an occupied cell within reach: cells 4 tx - 1 .. 4 tx + 4 of bands ty - 1 .. ty + 1
        for (int ty = wv; ty < tiles_y; ty += nwv) {
            if (lane < CELLBM_WORDS) {
                u64 m = cellbm ? 0ull : ~0ull;
                if (cellbm)
                    for (int b = ty - 1; b <= ty + 1; b++)
                        if (b >= 0 && b < bm_bands) m |= cellbm[((size_t)g * bm_bands + b) * CELLBM_WORDS + lane];
                rowm[wv][lane] = m;
            }
            __builtin_amdgcn_wave_barrier();
            int n = 0;
            for (int half = 0; half < 2; half++) {
                int tx = lane + 64 * half;
                bool on = tx < tiles_x && dct_cells(rowm[wv], 4 * tx - 1) != 0;
                u64 bal = __ballot(on);
                if (lane == 0) band_mask[ty][half] = bal;
                n += __popcll(bal);
            }
            if (lane == 0) band_off[ty] = n;
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();
        if (wv == 0) { // exclusive scan of the band counts: 8 bands per lane, then across the wave
            int v[DCT_MAXBANDS / 64], loc = 0;
            for (int k = 0; k < DCT_MAXBANDS / 64; k++) {
                int b = lane * (DCT_MAXBANDS / 64) + k;
                v[k] = b < tiles_y ? band_off[b] : 0;
                loc += v[k];
            }
            int inc = loc;
            for (int off = 1; off < 64; off <<= 1) {
                int t = __shfl_up(inc, off);
                if (lane >= off) inc += t;
            }
            int run = inc - loc;
            for (int k = 0; k < DCT_MAXBANDS / 64; k++) {
                int b = lane * (DCT_MAXBANDS / 64) + k;
                if (b < tiles_y) band_off[b] = run;
                run += v[k];
            }
            if (lane == 63) { band_off[DCT_MAXBANDS] = inc; counters[g * C_COUNT + C_NTILES] = inc < tile_cap ? inc : tile_cap; }
        }
        __syncthreads();
        int *tl = tile_list + (size_t)g * tile_cap;
        for (int ty = wv; ty < tiles_y; ty += nwv) {
            int o = band_off[ty];
            for (int half = 0; half < 2; half++) {
                u64 bal = band_mask[ty][half];
                if ((bal >> lane) & 1ull) {
                    int k = o + __popcll(bal & ((1ull << lane) - 1ull));
                    if (k < tile_cap) tl[k] = (ty << 16) | (lane + 64 * half);
                }
                o += __popcll(bal);
            }
        }
    }
    // ---- background of the outputs: the dilated, equalised image is lut[0] wherever nothing is within reach
    const unsigned z = lut ? lut[g * 256] : 0u;
    const int wq = LFD_WQ(w);
    const size_t BW = (size_t)h * wq;
    u64 *pe = equb + (size_t)g * BW, *pc = cand + (size_t)g * BW, *ps = strong + (size_t)g * BW;
    const size_t t0 = (size_t)fpart * DCT_THREADS + threadIdx.x, tstep = (size_t)fparts * DCT_THREADS;
    if (z == 0 && (BW & 1) == 0) {
        const uint4 zero = make_uint4(0, 0, 0, 0);
        for (size_t i = t0; i < BW / 2; i += tstep) {
            ((uint4 *)pe)[i] = zero; ((uint4 *)pc)[i] = zero; ((uint4 *)ps)[i] = zero;
        }
    } else {
        for (size_t i = t0; i < BW; i += tstep) {
            pe[i] = z ? valid_mask((int)(i % wq), w) : 0ull;
            pc[i] = 0ull; ps[i] = 0ull;
        }
    }
    if (equ) { // stage image requested: its background too
        const unsigned zz = z * 0x01010101u;
        uint4 *pq = (uint4 *)(equ + (size_t)g * h * w);
        for (size_t i = t0; i < (size_t)h * w / 16; i += tstep) pq[i] = make_uint4(zz, zz, zz, zz);
    }
}

// PROF: developer build with s_memtime stamps at the stage boundaries (LFDMI_DC_PROFILE=1, tools/dc_profile.py): per frame the
// cycles all waves spent in 0 input wait + staging, 1 masks, 2 horizontal max, 3 vertical max + LUT, 4 borders + equ bits,
// 5 Sobel, 6 NMS + stores, and (slot 7) the number of tiles that ran the stages.  The shipped kernel is the PROF = false one.
// KH, KW > 0: the structuring element's size as compile-time constants (the defaults of the two passes, 4 x 4 and 9 x 9, are
// instantiated: unrolled running maxima by doubling); 0, 0: any size, run-time loops.
#ifndef DCT_MIN_WAVES
#define DCT_MIN_WAVES 5
#endif
template <bool PROF, int KH, int KW>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(DCT_MIN_WAVES, 8)))
k_dilate_canny_t(const uint8_t *src, uint8_t *equ, u64 *equb, u64 *cand, u64 *strong, const uint8_t *lut, int h, int w,
                 int kh, int kw, int low, int high, const int *active, int nc, int parts, const int *tile_list, int tile_cap,
                 const int *counters, long long *prof, const int *perm) {
    int L = blockIdx.x, xcd = L & 7, jb_ = L >> 3;
    int g = (jb_ / parts) * 8 + xcd; // frame = 8 * (j / parts) + (block & 7): a frame's tiles behind one L2
    if (g >= nc) return;
    // Workgroups go to the eight XCDs round-robin, so frame slot f works on XCD f % 8.  A dim pass only works on the frames the
    // bright pass left undecided, whatever slots those are (every other slot on the benchmark's frames: four XCDs idle): `perm`
    // lists the active slots first, so the i-th ACTIVE frame works on XCD i % 8.
    if (perm) g = perm[g];
    if (active && !active[g]) return;
    const int part = jb_ - (jb_ / parts) * parts;
    const int ntl = counters[g * C_COUNT + C_NTILES];
    const int t_begin = (int)((long long)ntl * part / parts), t_end = (int)((long long)ntl * (part + 1) / parts);
    if (t_begin >= t_end) return;
    const int *tl = tile_list + (size_t)g * tile_cap;
    extern __shared__ __attribute__((aligned(16))) uint8_t smw[];
    const int PH = DCW_PH, MH = DCW_MH, NWD = DCW_NWD, TS = DCW_TS, TSW = DCW_TS / 4;
    const int IH = PH + kh - 1;
    const int MGB = MH * CANNY_MW * 2;                 // magnitude plane: ushort (m << 2 | sector)
    const int tin_bytes = IH * TS > MGB ? IH * TS : MGB;
    uint8_t *tin = smw;                                // IH x 112; later the magnitude plane
    uint32_t *tmp = (uint32_t *)(smw + tin_bytes);      // IH x 18 words of horizontal maxima
    uint8_t *px = (uint8_t *)(tmp + IH * NWD);         // PH x 112
    unsigned short *mg = (unsigned short *)smw;
    __shared__ uint8_t slut[256];
    __shared__ uint32_t Pm[8];                         // input piece column p: rows holding a non-zero byte
    __shared__ u64 rowc[DCW_TH], rows[DCW_TH];         // NMS output bit rows of the tile
    __shared__ u64 rowe[DCW_TH];                       // != 0 bits of the dilated, equalised tile
    const int lane = threadIdx.x;
    const int ay = kh / 2, ax = kw / 2;
    const size_t N = (size_t)h * w;
    const uint8_t *s = src + (size_t)g * N;
    const int wq = LFD_WQ(w);
    uint8_t *d = equ ? equ + (size_t)g * N : nullptr;
    const int npieces = IH * 6;                        // <= 192: three 16-byte pieces per lane
    int iy[3], wx[3];
    bool has[3];
#pragma unroll
    for (int p = 0; p < 3; p++) {
        int idx = lane + 64 * p;
        iy[p] = idx / 6;
        wx[p] = idx - iy[p] * 6;
        has[p] = idx < npieces;
    }
#pragma unroll
    for (int p = 0; p < 4; p++) slut[lane + 64 * p] = lut ? lut[g * 256 + lane + 64 * p] : (uint8_t)(lane + 64 * p);
    if (lane < 8) Pm[lane] = 0u;
    __syncthreads();
    uint32_t zw = slut[0]; // background of the dilated, equalised image
    zw |= zw << 8; zw |= zw << 16;
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    const uint32_t mPH = (1u << PH) - 1, mMH = (1u << MH) - 1, mIH = IH >= 32 ? ~0u : ((1u << IH) - 1);
    auto load_tile = [&](int code, uint4 *v) {
        const int ly0 = (code >> 16) * DCW_TH - 2 - ay, xl = (code & 0xffff) * CANNY_TW - CANNY_HALO;
#pragma unroll
        for (int p = 0; p < 3; p++) {
            v[p] = make_uint4(0, 0, 0, 0); // out-of-image samples are ignored by a dilation
            int gy = ly0 + iy[p], gx = xl + 16 * wx[p];
            if (has[p] && gy >= 0 && gy < h && gx >= 0 && gx < w) v[p] = *(const uint4 *)(s + (size_t)gy * w + gx);
        }
    };
    uint4 v[3];
    long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pt0 = 0;
#define DCW_STAMP(k) do { if (PROF) { long long t_ = (long long)__builtin_amdgcn_s_memtime(); pacc[k] += t_ - pt0; pt0 = t_; } } while (0)
    // the wave's share of the list is fetched up front, one entry per lane (a share is 3-12 tiles on SDSS frames, at most 64 on
    // 4096 x 4096 ones): the next tile's input can be requested without first waiting for its list entry to arrive
    for (int tb = t_begin; tb < t_end; tb += 64) {
    const int cnt = min(64, t_end - tb);
    const int codes = lane < cnt ? tl[tb + lane] : 0;
    int code = __builtin_amdgcn_readlane(codes, 0);
    load_tile(code, v);
    for (int tk = 0; tk < cnt; tk++) {
        if (PROF) pt0 = (long long)__builtin_amdgcn_s_memtime();
        const int tx = code & 0xffff, x0 = tx * CANNY_TW, y0 = (code >> 16) * DCW_TH;
        uint32_t anyv = 0;
#pragma unroll
        for (int p = 0; p < 3; p++)
            if (has[p]) {
                uint32_t nzv = v[p].x | v[p].y | v[p].z | v[p].w;
                if (nzv) atomicOr(&Pm[wx[p]], 1u << iy[p]);
                anyv |= nzv;
                *(uint4 *)(tin + iy[p] * TS + wx[p] * 16) = v[p];
            }
        if (tk + 1 < cnt) { // next tile's input: in flight while this one is processed
            code = __builtin_amdgcn_readlane(codes, tk + 1);
            load_tile(code, v);
        }
        if (__ballot(anyv != 0) == 0ull) continue; // cells marked, bytes clear after all: the background is already in place
        constexpr int DCW_KHC = KH, DCW_KWC = KW;
#include "k_dcw_tile.inc"
        if (PROF) pacc[7] += 1;
    }
    }
#undef DCW_STAMP
    if (PROF && prof && lane == 0)
        for (int k = 0; k < 8; k++) atomicAdd((unsigned long long *)&prof[(size_t)g * 8 + k], (unsigned long long)pacc[k]);
}

// generic widths

__global__ void __launch_bounds__(256)
k_canny_nms(const uint8_t *img, u64 *cand, u64 *strong, int h, int w, int low, int high,
            const int *active) {
    int g = blockIdx.z;
    if (active && !active[g]) return;
    const int PW = CANNY_TW + 4, PH = CANNY_TH + 4, MW = CANNY_TW + 2, MH = CANNY_TH + 2;
    __shared__ uint8_t px[PH * PW];
    __shared__ int mg[MH * MW];       // L1 gradient magnitude
    __shared__ int dxy[MH * MW];      // Sobel dx (low 16 bits) | dy (high 16 bits)
    __shared__ int rowflag[PH];       // staged pixel row holds a non-zero byte
    int x0 = blockIdx.x * CANNY_TW, y0 = blockIdx.y * CANNY_TH;
    const uint8_t *s = img + (size_t)g * h * w;
    if (threadIdx.x < PH) rowflag[threadIdx.x] = 0;
    __syncthreads();
    uint32_t any = 0;
    if ((w & 3) == 0) {
        // interior columns as aligned dwords (16 per row), the 2-px halo columns as bytes
        for (int idx = threadIdx.x; idx < PH * 16; idx += 256) {
            int ty = idx >> 4, wx = idx & 15;
            int gy = min(max(y0 - 2 + ty, 0), h - 1), gx = x0 + 4 * wx;
            uint32_t v;
            if (gx + 3 < w) v = *(const uint32_t *)(s + (size_t)gy * w + gx);
            else { uint32_t e = s[(size_t)gy * w + w - 1]; v = e | (e << 8) | (e << 16) | (e << 24); }
            if (v) rowflag[ty] = 1;
            any |= v;
            uint8_t *p = px + ty * PW + 2 + 4 * wx;
            p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24);
        }
        for (int idx = threadIdx.x; idx < PH * 4; idx += 256) {
            int ty = idx >> 2, k = idx & 3;
            int tx = k < 2 ? k : CANNY_TW + k; // 0,1 | 66,67
            int gy = min(max(y0 - 2 + ty, 0), h - 1), gx = min(max(x0 - 2 + tx, 0), w - 1);
            uint8_t v = s[(size_t)gy * w + gx];
            if (v) rowflag[ty] = 1;
            any |= v;
            px[ty * PW + tx] = v;
        }
    } else {
        for (int idx = threadIdx.x; idx < PH * PW; idx += 256) {
            int ty = idx / PW, tx = idx - ty * PW;
            int gy = min(max(y0 - 2 + ty, 0), h - 1), gx = min(max(x0 - 2 + tx, 0), w - 1);
            uint8_t v = s[(size_t)gy * w + gx];
            if (v) rowflag[ty] = 1;
            any |= v;
            px[idx] = v;
        }
    }
    int wq = LFD_WQ(w);
    // an all-zero tile (sky) has zero gradient everywhere: no candidate when low >= 0
    if (!__syncthreads_or(any != 0) && low >= 0) {
        if (threadIdx.x < CANNY_TH && y0 + threadIdx.x < h) {
            size_t o = (size_t)g * h * wq + (size_t)(y0 + threadIdx.x) * wq + blockIdx.x;
            cand[o] = 0ull;
            strong[o] = 0ull;
        }
        return;
    }
    // Sobel + magnitude once per pixel of the tile and its 1-px ring
    for (int idx = threadIdx.x; idx < MH * MW; idx += 256) {
        int my = idx / MW, mx = idx - my * MW;
        int gy = y0 - 1 + my, gx = x0 - 1 + mx;
        int m = 0, d = 0;
        if ((rowflag[my] | rowflag[my + 1] | rowflag[my + 2]) && gy >= 0 && gy < h && gx >= 0 && gx < w) {
            const uint8_t *p = px + my * PW + mx; // top-left of the 3x3 window
            int dx = (p[2] - p[0]) + 2 * (p[PW + 2] - p[PW]) + (p[2 * PW + 2] - p[2 * PW]);
            int dy = (p[2 * PW] - p[0]) + 2 * (p[2 * PW + 1] - p[1]) + (p[2 * PW + 2] - p[2]);
            m = abs(dx) + abs(dy);
            d = (dx & 0xffff) | (dy << 16);
        }
        mg[idx] = m;
        dxy[idx] = d;
    }
    __syncthreads();
    int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int oy = wv; oy < CANNY_TH; oy += 4) {
        int gy = y0 + oy, gx = x0 + lane;
        bool keep = false, str = false;
        if (gy < h && gx < w) {
            const int *mc = mg + (oy + 1) * MW + (lane + 1);
            int m = mc[0];
            if (m > low) {
                int d = dxy[(oy + 1) * MW + (lane + 1)];
                int xs = (int)(short)(d & 0xffff), ys = d >> 16;
                int ax = abs(xs), ay = abs(ys) << 15;
                int tg22x = ax * 13573;
                if (ay < tg22x) {
                    keep = (m > mc[-1]) && (m >= mc[1]);
                } else {
                    int tg67x = tg22x + (ax << 16);
                    if (ay > tg67x) {
                        keep = (m > mc[-MW]) && (m >= mc[MW]);
                    } else {
                        int sgn = ((xs ^ ys) < 0) ? -1 : 1;
                        keep = (m > mc[-MW - sgn]) && (m > mc[MW + sgn]);
                    }
                }
                str = keep && (m > high);
            }
        }
        u64 bc = __ballot(keep), bs = __ballot(str);
        if (lane == 0 && gy < h) {
            size_t o = (size_t)g * h * wq + (size_t)gy * wq + blockIdx.x;
            cand[o] = bc;
            strong[o] = bs;
        }
    }
}
