// k_ccl.h -- connectivity on bit rows: hysteresis (Canny's second half) and the topology
// findContours needs, without border following.
//
// cv2.findContours(RETR_LIST) (processfield.py:241-246) returns one border per adjacent pair
// (8-connected 1-component S1, 4-connected 0-component S2).  fit_minAreaRect only uses each
// border's point SET (convex hull -> minAreaRect), and
//   * the hull of an outer border == the hull of all pixels of S1,
//   * a hole border == the pixels of the surrounding component A that are 4-adjacent to the
//     hole B, where A is the component of the pixel right above B's raster-first pixel,
// so labelling runs of the bit rows (union-find over runs, root = raster-first run) replaces the
// sequential Suzuki-Abe trace.  Work is proportional to the number of runs (~ edge pixels +
// rows), not to the image area.
//
// Runs carry COMPACT ids: the scan kernels (k_scan_count / k_scan_bases / k_scan_write) store, per 64-bit word, how many runs start before it
// (exclusive scan in raster order), so the run that holds pixel (y, x) is
//   scan[word] + popcount(start bits of the word at columns <= x) - 1
// -- an O(1) lookup -- and all label arrays have one entry per run (a few thousand per frame,
// cache resident) instead of one per pixel (12 MB per array, every lookup an HBM miss).
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

// The candidate-pass scan (k_scan_write below) also builds both work lists of a frame:
//   fg: words holding candidate bits (edge runs are a subset of candidate runs);
//   bg: words where a 0-run of the edge image can start or a vertical 0-0 contact stretch can
//       begin -- an edge bit in this word or at the end of the previous word, in this row or the
//       row above -- or the first word of a row.  Candidate bits are a superset of edge bits, so
//       testing them gives a (harmless) superset of the words the hole kernels need.
// every run kernel walks its frame's work list with a fixed grid
#define LFD_WORDLIST_LOOP(cidx_)                                                          \
    int g = blockIdx.y;                                                                   \
    if (slot_off(active, counters, g)) return;                                            \
    const int wq = LFD_WQ(w);                                                             \
    const int nwork_ = counters[g * C_COUNT + (cidx_)];                                   \
    const int *wl_ = wlist + (size_t)g * h * wq;                                          \
    for (int it_ = blockIdx.x * 256 + threadIdx.x; it_ < nwork_; it_ += gridDim.x * 256)

#define WORDLIST_BLOCKS 48
#define SCAN_THREADS 1024

// scan[word] = number of `val`-runs that start in earlier words of the frame (raster order);
// counters[cidx] = total number of runs.  Three short, wide kernels (a single 1024-thread workgroup per frame
// reading its frame twice was bound by its own load latency): counting and writing parallelise over 64-word
// segments, only the scan of the per-segment counts is per frame.
#define SCAN_MAX_SEG 4096
//   k_scan_count: per 64-word segment (one wave each) the run starts and the work-list entries -> segcnt
//   k_scan_bases: per frame, exclusive scan of the segment counts in place, totals -> counters
//   k_scan_write: per segment, the per-word scan values, the work-list entries (raster order), the clear
#define SCANW_WAVES 4 // waves per workgroup
#define SCAN_SEGS 8   // consecutive segments per wave: all their loads are issued together (a wave per 512 bytes lived on
                      // launch overhead and on its own load latency: 1 M waves per launch on 4096 x 4096 frames)
// Per segment k of the wave (words ((seg0 + k) << 6) + lane): run starts of the lane's word, "holds a candidate bit" (fg work
// list) and "a 0-run can start or join here" (bg work list).  The word to the left comes from the neighbouring lane, the
// first lane's from the previous segment's last lane.
__device__ __forceinline__ void scan_words(const u64 *b, int seg0, int nseg, int nw, int wq, int val, int W, bool lists, int lane,
                                           int *c, bool *tf, bool *tb) {
    u64 cur[SCAN_SEGS], up[SCAN_SEGS];
    int yy[SCAN_SEGS], qq[SCAN_SEGS];
    bool in[SCAN_SEGS];
    const float inv = 1.0f / (float)wq;
#pragma unroll
    for (int k = 0; k < SCAN_SEGS; k++) {
        const int i = ((seg0 + k) << 6) + lane;
        int y = (int)((float)i * inv), q = i - y * wq; // i < 2^24: the quotient is off by one at most
        if (q < 0) { y--; q += wq; } else if (q >= wq) { y++; q -= wq; }
        yy[k] = y; qq[k] = q;
        in[k] = seg0 + k < nseg && i < nw;
        cur[k] = in[k] ? b[i] : 0ull;
        up[k] = (in[k] && lists && y > 0) ? b[i - wq] : 0ull;
    }
    u64 first_prev = 0ull, first_upprev = 0ull;
    {
        const int i = seg0 << 6;
        if (lane == 0 && i > 0 && i < nw) {
            first_prev = b[i - 1];
            if (lists && i - wq - 1 >= 0) first_upprev = b[i - wq - 1];
        }
    }
#pragma unroll
    for (int k = 0; k < SCAN_SEGS; k++) {
        u64 prev = __shfl_up(cur[k], 1), upprev = __shfl_up(up[k], 1);
        {
            auto last = [](u64 v) { // lane 63's value, read by every lane
                return (u64)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, 63) |
                       ((u64)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), 63) << 32);
            };
            const u64 lp = k > 0 ? last(cur[k > 0 ? k - 1 : 0]) : first_prev, lu = k > 0 ? last(up[k > 0 ? k - 1 : 0]) : first_upprev;
            if (lane == 0) { prev = lp; upprev = lu; }
        }
        const int q = qq[k], y = yy[k];
        u64 cc = (val ? cur[k] : ~cur[k]) & valid_mask(q, W);
        const u64 pm = q > 0 ? ((val ? prev : ~prev) >> 63) : 0ull;
        c[k] = in[k] ? __popcll(cc & ~((cc << 1) | pm)) : 0;
        tf[k] = false; tb[k] = false;
        if (lists && in[k]) {
            u64 m = cur[k];
            tf[k] = m != 0;
            if (q > 0) m |= prev >> 63;
            if (y > 0) { m |= up[k]; if (q > 0) m |= upprev >> 63; }
            tb[k] = (m != 0) || (q == 0);
        }
    }
}


__global__ void __launch_bounds__(64 * SCANW_WAVES)
k_scan_count(const u64 *bits, int val, int4 *segcnt, int h, int w, int lists, const int *active) {
    int g = blockIdx.y;
    if (active && !active[g]) return;
    const int wq = LFD_WQ(w), nw = h * wq, nseg = (nw + 63) >> 6;
    const int seg0 = (blockIdx.x * SCANW_WAVES + (threadIdx.x >> 6)) * SCAN_SEGS, lane = threadIdx.x & 63;
    if (seg0 >= nseg) return;
    int c[SCAN_SEGS]; bool tf[SCAN_SEGS], tb[SCAN_SEGS];
    scan_words(bits + (size_t)g * nw, seg0, nseg, nw, wq, val, w, lists != 0, lane, c, tf, tb);
#pragma unroll
    for (int k = 0; k < SCAN_SEGS; k++) {
        if (seg0 + k >= nseg) break;
        int cs = c[k];
        for (int off = 32; off > 0; off >>= 1) cs += __shfl_down(cs, off);
        int nf = __popcll(__ballot(tf[k])), nb = __popcll(__ballot(tb[k]));
        if (lane == 0) segcnt[(size_t)g * SCAN_MAX_SEG + seg0 + k] = make_int4(cs, nf, nb, 0);
    }
}

__global__ void __launch_bounds__(SCAN_THREADS)
k_scan_bases(int4 *segcnt, int *counters, int cidx, int h, int w, int run_cap, int lists, const int *active) {
    int g = blockIdx.x;
    if (active && !active[g]) return;
    const int nseg = (h * LFD_WQ(w) + 63) >> 6;
    int4 *sg = segcnt + (size_t)g * SCAN_MAX_SEG;
    const int per = SCAN_MAX_SEG / SCAN_THREADS; // 4 consecutive segments per thread
    int t0 = threadIdx.x * per, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int4 v[SCAN_MAX_SEG / SCAN_THREADS];
    int lc = 0, lf = 0, lb = 0;
    for (int k = 0; k < per; k++) {
        v[k] = t0 + k < nseg ? sg[t0 + k] : make_int4(0, 0, 0, 0);
        lc += v[k].x; lf += v[k].y; lb += v[k].z;
    }
    int ic = lc, if_ = lf, ib = lb;
    for (int off = 1; off < 64; off <<= 1) {
        int a = __shfl_up(ic, off), b2 = __shfl_up(if_, off), c2 = __shfl_up(ib, off);
        if (lane >= off) { ic += a; if_ += b2; ib += c2; }
    }
    __shared__ int wt[SCAN_THREADS / 64][3];
    if (lane == 63) { wt[wv][0] = ic; wt[wv][1] = if_; wt[wv][2] = ib; }
    __syncthreads();
    int bc = 0, bf = 0, bb = 0;
    for (int k = 0; k < wv; k++) { bc += wt[k][0]; bf += wt[k][1]; bb += wt[k][2]; }
    int rc = bc + ic - lc, rf = bf + if_ - lf, rb = bb + ib - lb;
    for (int k = 0; k < per; k++)
        if (t0 + k < nseg) {
            sg[t0 + k] = make_int4(rc, rf, rb, 0);
            rc += v[k].x; rf += v[k].y; rb += v[k].z;
        }
    if (threadIdx.x == SCAN_THREADS - 1) {
        counters[g * C_COUNT + cidx] = rc;
        if (rc > run_cap) counters[g * C_COUNT + C_OVERFLOW] = 1;
        if (lists) { counters[g * C_COUNT + C_NFGW] = rf; counters[g * C_COUNT + C_NBGW] = rb; }
    }
}

__global__ void __launch_bounds__(64 * SCANW_WAVES)
k_scan_write(const u64 *bits, int val, const int4 *segcnt, int *scan, int h, int w, int *wl_fg, int *wl_bg, u64 *clear,
             const int *active) {
    int g = blockIdx.y;
    if (active && !active[g]) return;
    const int wq = LFD_WQ(w), nw = h * wq, nseg = (nw + 63) >> 6;
    const int seg0 = (blockIdx.x * SCANW_WAVES + (threadIdx.x >> 6)) * SCAN_SEGS, lane = threadIdx.x & 63;
    if (seg0 >= nseg) return;
    int c[SCAN_SEGS]; bool tf[SCAN_SEGS], tb[SCAN_SEGS];
    scan_words(bits + (size_t)g * nw, seg0, nseg, nw, wq, val, w, wl_fg != nullptr, lane, c, tf, tb);
    int4 base[SCAN_SEGS];
#pragma unroll
    for (int k = 0; k < SCAN_SEGS; k++) base[k] = seg0 + k < nseg ? segcnt[(size_t)g * SCAN_MAX_SEG + seg0 + k] : make_int4(0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < SCAN_SEGS; k++) {
        if (seg0 + k >= nseg) break;
        const int i = ((seg0 + k) << 6) + lane;
        int incl = c[k];
        for (int off = 1; off < 64; off <<= 1) {
            int t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        if (wl_fg) {
            u64 bf = __ballot(tf[k]), bb = __ballot(tb[k]), lt = (1ull << lane) - 1ull;
            if (tf[k]) wl_fg[(size_t)g * nw + base[k].y + __popcll(bf & lt)] = i;
            if (tb[k]) wl_bg[(size_t)g * nw + base[k].z + __popcll(bb & lt)] = i;
        }
        if (i < nw) {
            scan[(size_t)g * nw + i] = base[k].x + incl - c[k];
            if (clear) clear[(size_t)g * nw + i] = 0ull;
        }
    }
}

// The three scan kernels in one launch (round 3): a workgroup counts its 32 segments, publishes its totals, adds up the
// totals of the frame's workgroups before it (a decoupled look-back: it waits for their COUNTS only, which they publish
// without waiting for anybody, so a workgroup is held up by at most the count phase of a neighbour that started late), and
// writes the per-word scan values and the work lists from the words it still holds -- the bit plane is read once instead of
// twice and two launches (~8 us of drain each) go.  `partial` holds one word per (frame slot, workgroup): totals + the
// launch's epoch as the "published" mark (every launch of a context has its own epoch; the host clears the words when the epoch wraps).  The wait is
// bounded (~0.5 ms; a neighbour's count phase is ~20 us): a workgroup that gives up flags the frame as overflowed (it is
// then run again in the worst-case workspace) and raises PASS_FLAG_SCAN_GAVEUP, on which the host goes back to the three
// launches for the rest of the context's life.  That happens when several PROCESSES share the GPU: the waiting workgroups
// of one hold the CU slots the other's not yet dispatched workgroups need, and the other way round (measured: four ranks on
// one device, every look-back ran into a -- then 150 ms -- bound); with one process per GPU the predecessors of a waiting
// workgroup are resident or done, because every XCD hands its workgroups out in index order and publishing waits for nobody.
#define SCAN_MAX_BLK (SCAN_MAX_SEG / (SCANW_WAVES * SCAN_SEGS))
#define PASS_FLAG_SCAN_GAVEUP 512 // pass_flags bit (beside k_frame.h's PASS_FLAG_GENERAL): the look-back gave up on this frame
__global__ void __launch_bounds__(64 * SCANW_WAVES)
k_scan_fused(const u64 *bits, int val, u64 *partial, int epoch /* 1 .. 2^22 - 1 */, int max_spin, int *pass_flags, int *scan, int h, int w, int *wl_fg, int *wl_bg, u64 *clear,
             int *counters, int cidx, int run_cap, const int *active) {
    const int g = blockIdx.y, bx = blockIdx.x;
    if (active && !active[g]) return;
    const int wq = LFD_WQ(w), nw = h * wq, nseg = (nw + 63) >> 6;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int seg0 = (bx * SCANW_WAVES + wv) * SCAN_SEGS;
    const bool lists = wl_fg != nullptr;
    int c[SCAN_SEGS]; bool tf[SCAN_SEGS], tb[SCAN_SEGS];
    scan_words(bits + (size_t)g * nw, seg0, nseg, nw, wq, val, w, lists, lane, c, tf, tb);
    int cs[SCAN_SEGS], nf[SCAN_SEGS], nb[SCAN_SEGS], incl[SCAN_SEGS];
    int wc = 0, wf = 0, wb = 0;
#pragma unroll
    for (int k = 0; k < SCAN_SEGS; k++) {
        int v = c[k];
        for (int off = 1; off < 64; off <<= 1) {
            int t = __shfl_up(v, off);
            if (lane >= off) v += t;
        }
        incl[k] = v;
        cs[k] = __builtin_amdgcn_readlane(v, 63);
        nf[k] = __popcll(__ballot(tf[k]));
        nb[k] = __popcll(__ballot(tb[k]));
        wc += cs[k]; wf += nf[k]; wb += nb[k];
    }
    __shared__ int wt[SCANW_WAVES][3];
    __shared__ int base_s[4]; // frame-wide totals before this workgroup: runs, fg words, bg words; [3]: look-back gave up
    if (lane == 0) { wt[wv][0] = wc; wt[wv][1] = wf; wt[wv][2] = wb; }
    __syncthreads();
    // one 64-bit word per workgroup carries totals and mark together -- [epoch:22][bg words:12][fg words:12][runs:18] -- so a
    // relaxed device-scope store / load pair is all the protocol needs (a release / acquire pair costs an L2 write-back per
    // workgroup and an invalidate per poll on this part: measured 3x the three-kernel scan)
    u64 *pg = partial + (size_t)g * SCAN_MAX_BLK;
    int tc = 0, tfw = 0, tbw = 0;
    for (int k = 0; k < SCANW_WAVES; k++) { tc += wt[k][0]; tfw += wt[k][1]; tbw += wt[k][2]; }
    if (threadIdx.x == 0)
        __hip_atomic_store(&pg[bx], ((u64)(unsigned)epoch << 42) | ((u64)tbw << 30) | ((u64)tfw << 18) | (u64)tc, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    if (wv == 0) {
        int bc = 0, bf = 0, bb = 0, bad = 0;
        for (int p0 = 0; p0 < bx; p0 += 64) {
            const int p = p0 + lane;
            if (p < bx) {
                bool ok = false;
                u64 v = 0;
                for (int spin = 0; spin < max_spin; spin++) {
                    v = __hip_atomic_load(&pg[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((int)(v >> 42) == epoch) { ok = true; break; }
                    __builtin_amdgcn_s_sleep(2);
                }
                if (ok) { bc += (int)(v & 0x3FFFFu); bf += (int)((v >> 18) & 0xFFFu); bb += (int)((v >> 30) & 0xFFFu); }
                else bad = 1;
            }
        }
        for (int off = 32; off > 0; off >>= 1) {
            bc += __shfl_xor(bc, off); bf += __shfl_xor(bf, off); bb += __shfl_xor(bb, off); bad |= __shfl_xor(bad, off);
        }
        if (lane == 0) { base_s[0] = bc; base_s[1] = bf; base_s[2] = bb; base_s[3] = bad; }
    }
    __syncthreads();
    int rc = base_s[0], rf = base_s[1], rb = base_s[2];
    for (int k = 0; k < wv; k++) { rc += wt[k][0]; rf += wt[k][1]; rb += wt[k][2]; }
    if (bx == (int)gridDim.x - 1 && threadIdx.x == 0) { // the frame's totals (the last workgroup: everything before it + its own)
        int *cnt = counters + (size_t)g * C_COUNT;
        const int total = base_s[0] + tc;
        cnt[cidx] = total;
        if (total > run_cap) cnt[C_OVERFLOW] = 1;
        if (lists) { cnt[C_NFGW] = base_s[1] + tfw; cnt[C_NBGW] = base_s[2] + tbw; }
    }
    if (base_s[3] && threadIdx.x == 0) {
        counters[(size_t)g * C_COUNT + C_OVERFLOW] = 1;
        atomicOr(&pass_flags[g], PASS_FLAG_SCAN_GAVEUP);
    }
    if (seg0 >= nseg) return;
#pragma unroll
    for (int k = 0; k < SCAN_SEGS; k++) {
        if (seg0 + k >= nseg) break;
        const int i = ((seg0 + k) << 6) + lane;
        if (lists) {
            u64 bfm = __ballot(tf[k]), bbm = __ballot(tb[k]), lt = (1ull << lane) - 1ull;
            if (tf[k]) wl_fg[(size_t)g * nw + rf + __popcll(bfm & lt)] = i;
            if (tb[k]) wl_bg[(size_t)g * nw + rb + __popcll(bbm & lt)] = i;
        }
        if (i < nw) {
            scan[(size_t)g * nw + i] = rc + incl[k] - c[k];
            if (clear) clear[(size_t)g * nw + i] = 0ull;
        }
        rc += cs[k]; rf += nf[k]; rb += nb[k];
    }
}

// id of the `val`-run holding pixel (y, x) (the pixel must have that value)
__device__ __forceinline__ int run_id(const int *scan_frame, const u64 *frame_bits, int y, int x, int val, int wq, int W) {
    int q = x >> 6, b = x & 63;
    u64 s = start_bits(frame_bits + (size_t)y * wq, q, val, W);
    u64 m = (b == 63) ? ~0ull : ((2ull << b) - 1ull);
    return scan_frame[y * wq + q] + __popcll(s & m) - 1;
}

// L[id] = id, YM[id] = ROW[id] = row, FL[id] = 0 for every run
__global__ void __launch_bounds__(256)
k_runs_init(const u64 *bits, int val, const int *scan, int *L, int *YM, int *FL, int *ROW, int h, int w, int run_cap,
            const int *wlist, const int *counters, int cidx, const int *active) {
    LFD_WORDLIST_LOOP(cidx) {
        int idx = wl_[it_];
        int y = idx / wq, q = idx - y * wq;
        const u64 *row = bits + (size_t)g * h * wq + (size_t)y * wq;
        int n = __popcll(start_bits(row, q, val, w));
        int id0 = scan[(size_t)g * h * wq + idx];
        size_t o = (size_t)g * run_cap;
        for (int k = 0; k < n && id0 + k < run_cap; k++) {
            L[o + id0 + k] = id0 + k;
            YM[o + id0 + k] = y;
            ROW[o + id0 + k] = y;
            FL[o + id0 + k] = 0;
        }
    }
}

// 8-connectivity between runs of 1-bits in rows y and y-1
__global__ void __launch_bounds__(256)
k_runs_merge8(const u64 *bits, const int *scan, int *L, int h, int w, int run_cap, const int *wlist, const int *counters,
              int cidx, const int *active) {
    LFD_WORDLIST_LOOP(cidx) {
    int idx = wl_[it_];
    int y = idx / wq, q = idx - y * wq;
    if (y == 0) continue;
    const u64 *fb = bits + (size_t)g * h * wq;
    const int *sf = scan + (size_t)g * h * wq;
    const u64 *row = fb + (size_t)y * wq;
    const u64 *up = row - wq;
    u64 c = row[q];
    if (!c) continue;
    u64 u = up[q];
    u64 uprev = q > 0 ? up[q - 1] : 0ull, unext = q + 1 < wq ? up[q + 1] : 0ull;
    u64 uL = (u << 1) | (uprev >> 63); // bit x set <=> up[x-1]
    u64 uR = (u >> 1) | (unext << 63); // bit x set <=> up[x+1]
    int *Lg = L + (size_t)g * run_cap;
    u64 v0 = c & u;
    v0 &= ~(v0 << 1); // first column of every vertical-contact stretch
    u64 vm = c & uL & ~u, vp = c & uR & ~u; // diagonal contacts not implied by a vertical one
    while (v0) {
        int b = __ffsll((long long)v0) - 1;
        v0 &= v0 - 1;
        int x = (q << 6) + b;
        uf_union(Lg, run_id(sf, fb, y, x, 1, wq, w), run_id(sf, fb, y - 1, x, 1, wq, w));
    }
    while (vm) {
        int b = __ffsll((long long)vm) - 1;
        vm &= vm - 1;
        int x = (q << 6) + b;
        uf_union(Lg, run_id(sf, fb, y, x, 1, wq, w), run_id(sf, fb, y - 1, x - 1, 1, wq, w));
    }
    while (vp) {
        int b = __ffsll((long long)vp) - 1;
        vp &= vp - 1;
        int x = (q << 6) + b;
        uf_union(Lg, run_id(sf, fb, y, x, 1, wq, w), run_id(sf, fb, y - 1, x + 1, 1, wq, w));
    }
    }
}

// 4-connectivity between runs of 0-bits in rows y and y-1
__global__ void __launch_bounds__(256)
k_runs_merge4_bg(const u64 *bits, const int *scan, int *L, int h, int w, int run_cap, const int *wlist,
                 const int *counters, int cidx, const int *active) {
    LFD_WORDLIST_LOOP(cidx) {
    int idx = wl_[it_];
    int y = idx / wq, q = idx - y * wq;
    if (y == 0) continue;
    const u64 *fb = bits + (size_t)g * h * wq;
    const int *sb = scan + (size_t)g * h * wq;
    const u64 *row = fb + (size_t)y * wq;
    const u64 *up = row - wq;
    u64 v = ~row[q] & ~up[q] & valid_mask(q, w);
    // first column of every stretch; a stretch continuing from the previous word (both rows 0
    // at the last column of that word) was already joined there
    u64 cont = 0;
    if (q > 0) cont = (~row[q - 1] & ~up[q - 1]) >> 63;
    u64 st = v & ~((v << 1) | cont);
    int *Lg = L + (size_t)g * run_cap;
    while (st) {
        int b = __ffsll((long long)st) - 1;
        st &= st - 1;
        int x = (q << 6) + b;
        int ia = run_id(sb, fb, y, x, 0, wq, w), ib = run_id(sb, fb, y - 1, x, 0, wq, w);
        // two runs that both start at column 0 touch the frame: each is flagged "outside" on its
        // own, joining them would only build a 1 489-link chain down the left image border
        bool a0 = (ia == sb[y * wq]) && !(row[0] & 1ull), b0 = (ib == sb[(y - 1) * wq]) && !(up[0] & 1ull);
        if (a0 && b0) continue;
        uf_union(Lg, ia, ib);
    }
    }
}

// Path-compress every run to its root; YM[root] = last row of the component (edge components
// only; holes get theirs in k_bg_extent -- the outside is one giant component and every one of
// its runs would hammer the same word); FL[root] = 1 if (val==1) any pixel of the run is set in
// `mark` (strong edge pixels), or (val==0) the run touches the image frame (outside).
__global__ void __launch_bounds__(256)
k_runs_flatten(const u64 *bits, int val, const u64 *mark, const int *scan, int *L, int *YM, int *FL, int h, int w,
               int run_cap, const int *wlist, const int *counters, int cidx, const int *active) {
    LFD_WORDLIST_LOOP(cidx) {
    int idx = wl_[it_];
    int y = idx / wq, q = idx - y * wq;
    const u64 *row = bits + (size_t)g * h * wq + (size_t)y * wq;
    u64 s = start_bits(row, q, val, w);
    int id = scan[(size_t)g * h * wq + idx];
    int *Lg = L + (size_t)g * run_cap, *YMg = YM + (size_t)g * run_cap, *FLg = FL + (size_t)g * run_cap;
    for (; s && id < run_cap; id++) {
        int b = __ffsll((long long)s) - 1;
        s &= s - 1;
        int xs = (q << 6) + b;
        int root = uf_find(Lg, id);
        if (root != id) Lg[id] = root;
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
k_bg_extent(const u64 *bits, const int *scan, const int *L, int *YM, const int *FL, int h, int w, int run_cap,
            const int *wlist, const int *counters, int cidx, const int *active) {
    LFD_WORDLIST_LOOP(cidx) {
    int idx = wl_[it_];
    int y = idx / wq, q = idx - y * wq;
    const u64 *row = bits + (size_t)g * h * wq + (size_t)y * wq;
    int n = __popcll(start_bits(row, q, 0, w));
    int id0 = scan[(size_t)g * h * wq + idx];
    size_t o = (size_t)g * run_cap;
    for (int k = 0; k < n && id0 + k < run_cap; k++) {
        int root = L[o + id0 + k];
        if (!FL[o + root]) atomicMax(&YM[o + root], y);
    }
    }
}

// hysteresis result: edge = candidate runs whose component holds a strong pixel
// (words without candidates are zero in `edge`: cleared by the caller)
__global__ void __launch_bounds__(256)
k_edge_from_cand(const u64 *cand, const int *scan, const int *L, const int *FL, u64 *edge, int h, int w, int run_cap,
                 const int *wlist, const int *counters, int cidx, const int *active) {
    LFD_WORDLIST_LOOP(cidx) {
    int idx = wl_[it_];
    int y = idx / wq, q = idx - y * wq;
    const u64 *fb = cand + (size_t)g * h * wq;
    const u64 *row = fb + (size_t)y * wq;
    const int *sf = scan + (size_t)g * h * wq;
    size_t o = (size_t)g * run_cap;
    u64 c = row[q], rem = c, res = 0;
    while (rem) {
        int b = __ffsll((long long)rem) - 1;
        u64 inv = ~(c >> b);
        int len = inv ? (__ffsll((long long)inv) - 1) : (64 - b);
        u64 seg = (len >= 64 ? ~0ull : ((1ull << len) - 1)) << b;
        int id = run_id(sf, fb, y, (q << 6) + b, 1, wq, w);
        if (id >= 0 && id < run_cap && FL[o + L[o + id]]) res |= seg;
        rem &= ~seg;
    }
    edge[(size_t)g * h * wq + (size_t)y * wq + q] = res;
    }
}
