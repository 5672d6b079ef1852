// k_ccl.h -- connectivity on bit rows: hysteresis (Canny's second half) and the topology
// findContours needs, without border following.
//
// cv2.findContours(RETR_LIST) (processfield.py:241-246) returns one border per adjacent pair
// (8-connected 1-component S1, 4-connected 0-component S2).  fit_minAreaRect only uses each
// border's point SET (convex hull -> minAreaRect), and
//   * the hull of an outer border == the hull of all pixels of S1,
//   * a hole border == the pixels of the surrounding component A that are 4-adjacent to the
//     hole B, where A is the component of the pixel right above B's raster-first pixel,
// so labelling runs of the bit rows (union-find over run starts, labels indexed by the pixel
// index of the run start, root = raster-first run) replaces the sequential Suzuki-Abe trace.
// Work is proportional to the number of runs (~ edge pixels + rows), not to the image area.
// One thread per ACTIVE 64-bit word of a bit row: k_collect_words compacts the few words that
// hold any work (a few % of a sky frame) into a per-frame list, so every lane of the run
// kernels has a word to chew on and their dependent, cache-missing label loads overlap
// (thread-per-word over the whole bit image left ~4 % of the lanes busy).
#pragma once
#include "common.h"

// run-start bits of word wq for runs of value `val`
__device__ __forceinline__ u64 start_bits(const u64 *row, int wq, int val, int W) {
    u64 c = row[wq];
    u64 vm = valid_mask(wq, W);
    if (!val) c = ~c;
    c &= vm;
    u64 prev_msb = 0;
    if (wq > 0) {
        u64 pc = row[wq - 1];
        if (!val) pc = ~pc;
        prev_msb = pc >> 63;
    }
    return c & ~((c << 1) | prev_msb);
}

// One pass over the candidate bit rows builds both work lists of a frame:
//   fg: words holding candidate bits (edge runs are a subset of candidate runs);
//   bg: words where a 0-run of the edge image can start or a vertical 0-0 contact stretch can
//       begin -- an edge bit in this word or at the end of the previous word, in this row or the
//       row above -- or the first word of a row.  Candidate bits are a superset of edge bits, so
//       testing them gives a (harmless) superset of the words the hole kernels need.
__global__ void __launch_bounds__(256)
k_collect_words(const u64 *cand, int *wl_fg, int *wl_bg, int *counters, int h, int w, const int *active) {
    int g = blockIdx.y;
    if (active && !active[g]) return;
    int wq = LFD_WQ(w);
    int idx = blockIdx.x * 256 + threadIdx.x;
    bool tf = false, tb = false;
    if (idx < h * wq) {
        const u64 *b = cand + (size_t)g * h * wq;
        int y = idx / wq, q = idx - y * wq;
        u64 m = b[idx];
        tf = m != 0;
        if (q > 0) m |= b[idx - 1] >> 63;
        if (y > 0) { m |= b[idx - wq]; if (q > 0) m |= b[idx - wq - 1] >> 63; }
        tb = (m != 0) || (q == 0);
    }
    int lane = lfd_lane();
    u64 bf = __ballot(tf), bb = __ballot(tb);
    int basef = 0, baseb = 0;
    if (lane == 0) {
        if (bf) basef = atomicAdd(&counters[g * C_COUNT + C_NFGW], __popcll(bf));
        if (bb) baseb = atomicAdd(&counters[g * C_COUNT + C_NBGW], __popcll(bb));
    }
    basef = __shfl(basef, 0);
    baseb = __shfl(baseb, 0);
    u64 lt = (1ull << lane) - 1ull;
    if (tf) wl_fg[(size_t)g * h * wq + basef + __popcll(bf & lt)] = idx;
    if (tb) wl_bg[(size_t)g * h * wq + baseb + __popcll(bb & lt)] = idx;
}

// every run kernel walks its frame's work list with a fixed grid
#define LFD_WORDLIST_LOOP(cidx_)                                                          \
    int g = blockIdx.y;                                                                   \
    if (active && !active[g]) return;                                                     \
    const int wq = LFD_WQ(w);                                                             \
    const int nwork_ = counters[g * C_COUNT + (cidx_)];                                   \
    const int *wl_ = wlist + (size_t)g * h * wq;                                          \
    for (int it_ = blockIdx.x * 256 + threadIdx.x; it_ < nwork_; it_ += gridDim.x * 256)

#define WORDLIST_BLOCKS 48

// L[p] = p, YM[p] = row, FL[p] = 0 for every run start p
__global__ void __launch_bounds__(256)
k_runs_init(const u64 *bits, int val, int *L, int *YM, int *FL, int h, int w, const int *wlist,
            const int *counters, int cidx, const int *active) {
    LFD_WORDLIST_LOOP(cidx) {
        int idx = wl_[it_];
        int y = idx / wq, q = idx - y * wq;
        const u64 *row = bits + (size_t)g * h * wq + (size_t)y * wq;
        u64 s = start_bits(row, q, val, w);
        size_t N = (size_t)h * w;
        while (s) {
            int b = __ffsll((long long)s) - 1;
            s &= s - 1;
            int p = y * w + (q << 6) + b;
            L[g * N + p] = p;
            YM[g * N + p] = y;
            FL[g * N + p] = 0;
        }
    }
}

// 8-connectivity between runs of 1-bits in rows y and y-1
__global__ void __launch_bounds__(256)
k_runs_merge8(const u64 *bits, int *L, int h, int w, const int *wlist, const int *counters, int cidx,
              const int *active) {
    LFD_WORDLIST_LOOP(cidx) {
    int idx = wl_[it_];
    int y = idx / wq, q = idx - y * wq;
    if (y == 0) continue;
    const u64 *row = bits + (size_t)g * h * wq + (size_t)y * wq;
    const u64 *up = row - wq;
    u64 c = row[q];
    if (!c) continue;
    u64 u = up[q];
    u64 uprev = q > 0 ? up[q - 1] : 0ull, unext = q + 1 < wq ? up[q + 1] : 0ull;
    u64 uL = (u << 1) | (uprev >> 63); // bit x set <=> up[x-1]
    u64 uR = (u >> 1) | (unext << 63); // bit x set <=> up[x+1]
    int *Lg = L + (size_t)g * h * w;
    u64 v0 = c & u;
    v0 &= ~(v0 << 1); // first column of every vertical-contact stretch
    u64 vm = c & uL & ~u, vp = c & uR & ~u; // diagonal contacts not implied by a vertical one
    vm &= ~(vm << 1);
    while (v0) {
        int b = __ffsll((long long)v0) - 1;
        v0 &= v0 - 1;
        int x = (q << 6) + b;
        uf_union(Lg, y * w + run_start(row, x, 1), (y - 1) * w + run_start(up, x, 1));
    }
    while (vm) {
        int b = __ffsll((long long)vm) - 1;
        vm &= vm - 1;
        int x = (q << 6) + b;
        uf_union(Lg, y * w + run_start(row, x, 1), (y - 1) * w + run_start(up, x - 1, 1));
    }
    while (vp) {
        int b = __ffsll((long long)vp) - 1;
        vp &= vp - 1;
        int x = (q << 6) + b;
        uf_union(Lg, y * w + run_start(row, x, 1), (y - 1) * w + run_start(up, x + 1, 1));
    }
    }
}

// 4-connectivity between runs of 0-bits in rows y and y-1
__global__ void __launch_bounds__(256)
k_runs_merge4_bg(const u64 *bits, int *L, int h, int w, const int *wlist, const int *counters, int cidx,
                 const int *active) {
    LFD_WORDLIST_LOOP(cidx) {
    int idx = wl_[it_];
    int y = idx / wq, q = idx - y * wq;
    if (y == 0) continue;
    const u64 *row = bits + (size_t)g * h * wq + (size_t)y * wq;
    const u64 *up = row - wq;
    u64 v = ~row[q] & ~up[q] & valid_mask(q, w);
    // first column of every stretch; a stretch continuing from the previous word (both rows 0
    // at the last column of that word) was already joined there
    u64 cont = 0;
    if (q > 0) cont = (~row[q - 1] & ~up[q - 1]) >> 63;
    u64 st = v & ~((v << 1) | cont);
    int *Lg = L + (size_t)g * h * w;
    while (st) {
        int b = __ffsll((long long)st) - 1;
        st &= st - 1;
        int x = (q << 6) + b;
        int sa = run_start(row, x, 0), sb = run_start(up, x, 0);
        // two runs that both start at column 0 touch the frame: each is flagged "outside" on its
        // own, joining them would only build a 1 489-link chain down the left image border
        if (sa == 0 && sb == 0) continue;
        uf_union(Lg, y * w + sa, (y - 1) * w + sb);
    }
    }
}

// Path-compress every run start to its root; YM[root] = last row of the component;
// FL[root] = 1 if (val==1) any pixel of the run is set in `mark` (strong edge pixels), or
// (val==0) the run touches the image frame (the 0-component is the outside).
__global__ void __launch_bounds__(256)
k_runs_flatten(const u64 *bits, int val, const u64 *mark, int *L, int *YM, int *FL, int h, int w,
               const int *wlist, const int *counters, int cidx, const int *active) {
    LFD_WORDLIST_LOOP(cidx) {
    int idx = wl_[it_];
    int y = idx / wq, q = idx - y * wq;
    const u64 *row = bits + (size_t)g * h * wq + (size_t)y * wq;
    u64 s = start_bits(row, q, val, w);
    size_t N = (size_t)h * w;
    int *Lg = L + g * N, *YMg = YM + g * N, *FLg = FL + g * N;
    while (s) {
        int b = __ffsll((long long)s) - 1;
        s &= s - 1;
        int xs = (q << 6) + b;
        int p = y * w + xs;
        int root = uf_find(Lg, p);
        if (root != p) Lg[p] = root;
        // last row of the component: needed for edge components here; for 0-components only
        // holes need it (k_bg_extent) -- the outside is one giant component and every one of
        // its runs would hammer the same word
        if (val) atomicMax(&YMg[root], y);
        int xe = run_end(row, xs, val, w);
        bool flag;
        if (val) {
            const u64 *mrow = mark + (size_t)g * h * wq + (size_t)y * wq;
            flag = false;
            for (int k = xs >> 6; k <= (xe >> 6) && !flag; k++) {
                u64 m = mrow[k];
                int lo = (k == (xs >> 6)) ? (xs & 63) : 0, hi = (k == (xe >> 6)) ? (xe & 63) : 63;
                u64 msk = (~0ull << lo) & (~0ull >> (63 - hi));
                flag = (m & msk) != 0;
            }
        } else {
            flag = (y == 0) || (y == h - 1) || (xs == 0) || (xe == w - 1);
        }
        if (flag) FLg[root] = 1;
    }
    }
}

// last row of every hole (0-component that does not touch the frame); runs after k_runs_flatten
__global__ void __launch_bounds__(256)
k_bg_extent(const u64 *bits, const int *L, int *YM, const int *FL, int h, int w, const int *wlist,
            const int *counters, int cidx, const int *active) {
    LFD_WORDLIST_LOOP(cidx) {
    int idx = wl_[it_];
    int y = idx / wq, q = idx - y * wq;
    const u64 *row = bits + (size_t)g * h * wq + (size_t)y * wq;
    u64 s = start_bits(row, q, 0, w);
    size_t N = (size_t)h * w;
    while (s) {
        int b = __ffsll((long long)s) - 1;
        s &= s - 1;
        int root = L[g * N + y * w + (q << 6) + b];
        if (!FL[g * N + root]) atomicMax(&YM[g * N + root], y);
    }
    }
}

// hysteresis result: edge = candidate runs whose component holds a strong pixel
__global__ void __launch_bounds__(256)
k_edge_from_cand(const u64 *cand, const int *L, const int *FL, u64 *edge, int h, int w,
                 const int *wlist, const int *counters, int cidx, const int *active) {
    // words without candidates are zero in `edge` (cleared by the caller)
    LFD_WORDLIST_LOOP(cidx) {
    int idx = wl_[it_];
    int y = idx / wq, q = idx - y * wq;
    const u64 *row = cand + (size_t)g * h * wq + (size_t)y * wq;
    size_t N = (size_t)h * w;
    const int *Lg = L + g * N, *FLg = FL + g * N;
    u64 c = row[q], rem = c, res = 0;
    while (rem) {
        int b = __ffsll((long long)rem) - 1;
        u64 inv = ~(c >> b);
        int len = inv ? (__ffsll((long long)inv) - 1) : (64 - b);
        u64 seg = (len >= 64 ? ~0ull : ((1ull << len) - 1)) << b;
        int start = (b == 0) ? run_start(row, q << 6, 1) : ((q << 6) + b);
        int root = Lg[y * w + start];
        if (FLg[root]) res |= seg;
        rem &= ~seg;
    }
    edge[(size_t)g * h * wq + (size_t)y * wq + q] = res;
    }
}
