// common.h -- device helpers shared by the lfdmi kernels (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned long long u64;

// per-slot counters (int32), zeroed before every pass
enum {
    C_NKEYS = 0,     // contour keys (fg components + holes)
    C_NSLOTS = 1,    // row-extent slots handed out
    C_NQUADS = 2,    // accepted rectangles
    C_NPIX_EQU = 3,  // entries of equ's Hough input list (pixel chunks, see k_pixlist)
    C_NPIX_BOX = 4,  // entries of box_img's list
    C_NPEAK_EQU = 5, // local maxima of the equ accumulator
    C_NPEAK_BOX = 6,
    C_OVERFLOW = 7,  // workspace overflow flag
    C_DETECT = 8,    // fit_minAreaRect's `detection`
    C_NBIG = 9,      // keys tall enough for the wave-per-key hull path
    C_NFGW = 10,     // bit-row words holding Canny candidates (work list of the edge-run kernels)
    C_NBGW = 11,     // bit-row words where a background run can start or join (work list of the hole kernels)
    C_NRUNF = 12,    // candidate runs in the frame (compact run ids 0 .. n-1)
    C_NRUNB = 13,    // background runs of the edge image
    C_NMED = 14,     // keys of medium height (one wave each, small LDS footprint)
    C_NNZ_EQU = 15,  // non-zero pixels of equ (180 Hough votes each)
    C_NNZ_BOX = 16,  // non-zero pixels of box_img
    C_NTILES = 17,   // 64 x 16 tiles of the pass image with anything in reach (work list of k_dilate_canny_t)
    C_NPIXB_EQU = 18, // entries of equ's class-B list (longer chunks for the angle slabs away from the horizontal, see k_pixlist)
    C_NPIXB_BOX = 19,
    C_COUNT = 20
};

#define LFD_WQ(w) (((w) + 63) >> 6)

__device__ __forceinline__ int lfd_lane() { return threadIdx.x & 63; }

// A slot is skipped by a kernel when the pass does not work on it (`active` mask) or when an earlier kernel of
// this pass found one of its tables too small (C_OVERFLOW): ids beyond a table's capacity must never be used
// as indices, so every kernel downstream of a capacity check leaves such a frame alone.  The host runs the
// frame again through the worst-case workspace (lfdmi.hip: spill).
__device__ __forceinline__ bool slot_off(const int *active, const int *counters, int g) {
    return (active && !active[g]) || counters[g * C_COUNT + C_OVERFLOW] != 0;
}

// mask of valid bits of word wq in a row of W pixels
__device__ __forceinline__ u64 valid_mask(int wq, int W) {
    int rem = W - (wq << 6);
    return rem >= 64 ? ~0ull : (rem <= 0 ? 0ull : ((1ull << rem) - 1));
}

// First column of the run (maximal stretch of equal bits == val) containing column x.
__device__ __forceinline__ int run_start(const u64 *row, int x, int val) {
    int wq = x >> 6, b = x & 63;
    u64 wv = row[wq];
    if (val) wv = ~wv; // 1-bits now mark "not in the run"
    u64 m = b ? (wv & (~0ull >> (64 - b))) : 0ull;
    for (;;) {
        if (m) return (wq << 6) + 64 - __clzll((long long)m);
        if (wq == 0) return 0;
        --wq;
        m = row[wq];
        if (val) m = ~m;
    }
}

// Last column of the run containing column x.
__device__ __forceinline__ int run_end(const u64 *row, int x, int val, int W) {
    int wq = x >> 6, b = x & 63, nwq = LFD_WQ(W);
    u64 wv = row[wq];
    if (val) wv = ~wv;
    u64 m = (b == 63) ? 0ull : (wv & (~0ull << (b + 1)));
    for (;;) {
        if (m) {
            int p = (wq << 6) + __ffsll((long long)m) - 1;
            return (p < W ? p : W) - 1;
        }
        if (wq == nwq - 1) return W - 1;
        ++wq;
        m = row[wq];
        if (val) m = ~m;
    }
}

__device__ __forceinline__ int get_bit(const u64 *row, int x) { return (int)((row[x >> 6] >> (x & 63)) & 1ull); }

// ---- lock-free union-find over an int label array (roots: L[x] == x, links go to smaller ids)
__device__ __forceinline__ int uf_find(const int *L, int x) {
    int p;
    while ((p = L[x]) != x) x = p;
    return x;
}

__device__ __forceinline__ void uf_union(int *L, int a, int b) {
    for (;;) {
        a = uf_find(L, a);
        b = uf_find(L, b);
        if (a == b) return;
        if (a < b) { int t = a; a = b; b = t; }
        int old = atomicMin(&L[a], b); // memory-side atomic: the true previous parent of a
        if (old == a) return;          // a was a root: linked
        a = old;                       // a had been linked meanwhile: merge its parent with b too
    }
}
