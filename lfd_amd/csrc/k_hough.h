// k_hough.h -- cv2.HoughLines(img, rho, theta, threshold) (processfield.py:370-371,488-489):
// pixel list from bit rows, the rho/theta vote accumulator in LDS, 4-neighbour local maxima,
// ordering by (votes desc, index asc), and the check_theta tail (processfield.py:36-150).
//
// Vote kernel layout: one workgroup owns a slab of <= 64 angles of one image; lane == angle,
// so every lane of a wave votes for the SAME pixel in a DIFFERENT accumulator row: no two
// lanes ever hit the same LDS word, cos/sin live in registers, the pixel coordinates are
// wave-uniform.  r = cvRound(j*tabCos[n] + i*tabSin[n]) is evaluated in float32 with separate
// roundings (no FMA) exactly as OpenCV's scalar loop, so the accumulator is integer-exact.
#pragma once
#include "common.h"
#include "../../include/lfdmi.h"

// bit rows -> lists of horizontal pixel CHUNKS, order irrelevant (votes commute):
//   entry = (len - 1) << 26 | y << 13 | x0,   1 <= len <= 64, the pixels (y, x0 .. x0 + len - 1) all set (one 64-pixel word).
// The Hough inputs are dilated blobs and filled rectangles: runs of 10-100 pixels, so a frame's list is many times shorter
// than its pixel count, and the vote kernel handles a chunk with two accumulator updates per angle instead of one per
// pixel (see k_hough_vote).  Image sides up to 8191 (13 bits).
// A chunk must span less than one rho bin at every angle that votes with it: (len - 1) |cos| / rho < 1.  That bound depends
// on the angle: 20 pixels at rho = 20 for the angles near 0 / 180 degrees, 45 for a slab of angles 64 .. 127 degrees.  So the
// runs are cut TWICE: list A with the chunk length every angle slab can take, list B (longer chunks, fewer entries) for the
// slabs whose angles all stay away from the horizontal (host: vote_classes); a slab votes with the list of its class.
#define PIXLIST_WORDS 8 // words per thread of k_pixlist
#define CHUNK_MAX 64 // a chunk lies inside one 64-pixel word (6-bit length field)
__device__ __forceinline__ uint32_t chunk_entry(int y, int x0, int len) {
    return ((uint32_t)(len - 1) << 26) | ((uint32_t)y << 13) | (uint32_t)x0;
}

// lists: [slot][class A / B][list_cap] per image; counters C_NPIX_* (class A) and C_NPIXB_* (class B); cm_b == 0: no class B
__global__ void __launch_bounds__(256)
k_pixlist(const u64 *bits0, const u64 *bits1, uint32_t *list0, uint32_t *list1, int *counters, int cm_a, int cm_b, int h, int w,
          size_t list_cap, int *accum_clear, int acc_n, size_t acc_stride, const int *active, int need_detect) {
    int g = blockIdx.y, im = blockIdx.z; // image 0: equ, image 1: box_img
    const u64 *bits = im ? bits1 : bits0;
    uint32_t *list = im ? list1 : list0;
    const int cidx = im ? C_NPIX_BOX : C_NPIX_EQU, cidx_b = im ? C_NPIXB_BOX : C_NPIXB_EQU, nnz_idx = im ? C_NNZ_BOX : C_NNZ_EQU;
    if (accum_clear) accum_clear += (size_t)im * acc_stride;
    if (slot_off(active, counters, g)) return;
    int *cnt = counters + g * C_COUNT;
    if (need_detect && !cnt[C_DETECT]) return;
    int wq = LFD_WQ(w);
    // zero this image's vote accumulator on the way (needed when the vote kernel merges split
    // lists with atomics; hipMemsetAsync's fill kernel is several times slower)
    if (accum_clear)
        for (int k = blockIdx.x * 256 + threadIdx.x; k < acc_n; k += gridDim.x * 256) accum_clear[(size_t)g * 2 * acc_stride + k] = 0;
    // PIXLIST_WORDS words per thread: a workgroup per 256 words spent its life on the two dependent loads in front
    // of the work (active flag, detection counter) -- 95 000 workgroups per launch, most of them with nothing to do
    const int nw = h * wq;
    uint32_t *lga = list + (size_t)g * 2 * list_cap, *lgb = lga + list_cap;
    const int lane = lfd_lane();
    u64 cw[PIXLIST_WORDS]; // all of a thread's words first: independent loads in flight together
#pragma unroll
    for (int it = 0; it < PIXLIST_WORDS; it++) {
        int idx = (blockIdx.x * PIXLIST_WORDS + it) * 256 + threadIdx.x;
        cw[it] = idx < nw ? bits[(size_t)g * nw + idx] : 0ull;
    }
    // Most words are empty and the rest are scattered: a wave first packs its non-zero words (up to 64 PIXLIST_WORDS of them)
    // into LDS and then cuts them into chunks with all lanes busy, instead of running the chunk loops once per 64 words
    // with one or two live lanes.
    __shared__ u64 sw[4][PIXLIST_WORDS * 64];
    __shared__ int si[4][PIXLIST_WORDS * 64];
    const int wv = threadIdx.x >> 6;
    int nloc = 0;
#pragma unroll
    for (int it = 0; it < PIXLIST_WORDS; it++) {
        const u64 bal = __ballot(cw[it] != 0);
        if (bal == 0ull) continue; // a wave of empty words (most of them)
        if (cw[it]) {
            int pos = nloc + __popcll(bal & ((1ull << lane) - 1ull));
            sw[wv][pos] = cw[it];
            si[wv][pos] = (blockIdx.x * PIXLIST_WORDS + it) * 256 + threadIdx.x;
        }
        nloc += __popcll(bal);
    }
    __builtin_amdgcn_wave_barrier();
    // pieces of a run of `len` pixels cut every cm: ceil(len / cm) without an integer division (len, cm <= 64)
    const float inv_a = (1.0f / (float)cm_a) * 1.000001f, inv_b = cm_b > 0 ? (1.0f / (float)cm_b) * 1.000001f : 0.0f;
    for (int kb = 0; kb < nloc; kb += 64) {
        u64 c = 0;
        int y = 0, q = 0;
        if (kb + lane < nloc) {
            int idx = si[wv][kb + lane];
            y = idx / wq; q = idx - y * wq;
            c = sw[wv][kb + lane] & valid_mask(q, w);
        }
        if (__ballot(c != 0) == 0ull) continue;
        // chunks of this word: every maximal stretch of set bits, cut every cm_a (list A) / cm_b (list B) pixels
        int na = 0, nb = 0;
        for (u64 r = c; r;) {
            int b = __ffsll((long long)r) - 1;
            u64 inv = ~(r >> b);
            int len = inv ? (__ffsll((long long)inv) - 1) : (64 - b);
            r &= ~((len >= 64 ? ~0ull : ((1ull << len) - 1)) << b);
            na += (int)((float)(len + cm_a - 1) * inv_a);
            nb += (int)((float)(len + cm_b - 1) * inv_b);
        }
        // wave-aggregated allocation: inclusive scans of the counts, one atomic per wave and list
        int incl_a = na, incl_b = nb, px = __popcll(c);
        for (int off = 1; off < 64; off <<= 1) {
            int ta = __shfl_up(incl_a, off), tb = __shfl_up(incl_b, off);
            if (lane >= off) { incl_a += ta; incl_b += tb; }
        }
        for (int off = 32; off > 0; off >>= 1) px += __shfl_down(px, off);
        const int total_a = __shfl(incl_a, 63), total_b = __shfl(incl_b, 63);
        int base_a = 0, base_b = 0;
        if (lane == 63 && total_a) base_a = atomicAdd(&cnt[cidx], total_a);
        if (lane == 63 && total_b) base_b = atomicAdd(&cnt[cidx_b], total_b);
        if (lane == 0 && px) atomicAdd(&cnt[nnz_idx], px);
        base_a = __shfl(base_a, 63);
        base_b = __shfl(base_b, 63);
        if ((size_t)base_a + (size_t)total_a > list_cap || (size_t)base_b + (size_t)total_b > list_cap) {
            // list full: the frame is run again through the worst-case workspace
            if (lane == 0) cnt[C_OVERFLOW] = 1;
            continue;
        }
        int oa = base_a + incl_a - na, ob = base_b + incl_b - nb;
        for (u64 r = c; r;) {
            int b = __ffsll((long long)r) - 1;
            u64 inv = ~(r >> b);
            int len = inv ? (__ffsll((long long)inv) - 1) : (64 - b);
            r &= ~((len >= 64 ? ~0ull : ((1ull << len) - 1)) << b);
            const int xr = (q << 6) + b;
            for (int x0 = xr, l = len; l > 0; x0 += cm_a, l -= cm_a) lga[oa++] = chunk_entry(y, x0, min(l, cm_a));
            if (cm_b > 0)
                for (int x0 = xr, l = len; l > 0; x0 += cm_b, l -= cm_b) lgb[ob++] = chunk_entry(y, x0, min(l, cm_b));
        }
    }
}

#define VOTE_THREADS 1024
// Bins an angle slab can reach: r = (x cos + y sin) / rho over the image rectangle covers only a part of the
// (numrho) accumulator rows -- 0 .. 127 of 356 for angles 0 - 63 deg of an SDSS frame -- so a workgroup's LDS slab
// holds rows lo .. hi only (host: vote_ranges): less than half the LDS, two to three workgroups per CU.
#define VOTE_MAX_SLABS 32
struct VoteRanges { int lo[VOTE_MAX_SLABS], hi[VOTE_MAX_SLABS]; unsigned cls_b; }; // centred bin index r (0 = rho 0), inclusive; bit sl of cls_b: slab sl votes with list B

// Accumulator layouts.  OpenCV indexes accum[(n+1)*(numrho+2) + r+1] ("base"); that value is
// still what orders equal-vote lines.  In memory the accumulator is kept TRANSPOSED,
// T[(r+1)*(numangle+2) + n+1], because the vote kernel's LDS slab is bin-major:
//   acc[r * AW + lane]   (AW = angles per workgroup, a power of two <= 64, lane = angle)
// so the LDS bank of a vote is lane % 32 whatever the data-dependent bin r is: the 32 lanes of
// a half-wave never collide (the angle-major layout lost 75 % of its LDS cycles to conflicts),
// and flushing a bin row is a coalesced store of AW consecutive angles.
//
// grid (nslabs * nsplit, n_images_per_slot, G).  With nsplit > 1 the pixel list of one image is
// cut into nsplit pieces whose slabs are merged with global atomic adds (accumulator zeroed
// beforehand); with nsplit == 1 the slab is stored directly.
template <int AWL> // log2 of the angles per workgroup
__global__ void __launch_bounds__(VOTE_THREADS)
k_hough_vote(const uint32_t *list0, const uint32_t *list1, const int *counters, const float *tab,
             int *accum, int numangle, int numrho, int nsplit, size_t list_cap, size_t acc_cap,
             const int *active, int need_detect, VoteRanges rng, int balance) {
    constexpr int aw_log2 = AWL;
    int g = blockIdx.z, im = blockIdx.y;
    int slab = blockIdx.x / nsplit, split = blockIdx.x - slab * nsplit;
    // everything this workgroup decides on, requested at once (seven independent loads: one round trip, not a chain of four)
    const int *cnt = counters + g * C_COUNT;
    const int on_ = active ? active[g] : 1, ovf_ = cnt[C_OVERFLOW], det_ = cnt[C_DETECT];
    const int na_e = cnt[C_NPIX_EQU], na_b = cnt[C_NPIX_BOX], nb_e = cnt[C_NPIXB_EQU], nb_b = cnt[C_NPIXB_BOX];
    if (!on_ || ovf_) return; // (slot_off)
    if (need_detect && !det_) return;
    const int cls = (slab < VOTE_MAX_SLABS) ? (int)((rng.cls_b >> slab) & 1u) : 0; // the chunk list this slab's angles may vote with
    if (balance) {
        // gridDim.y == 1 and the nsplit pieces of a slab are shared out between the TWO images in proportion to their list
        // lengths (equ's list is about three times box_img's): every workgroup of a frame then gets about the same number of
        // chunks, whatever XCD it lands on, instead of half the workgroups finishing in a third of the time of the others
        const int n0 = cls ? nb_e : na_e, n1 = cls ? nb_b : na_b;
        int s0 = (n0 + n1) > 0 ? (int)(((long long)nsplit * n0 + (n0 + n1) / 2) / (n0 + n1)) : nsplit / 2;
        s0 = max(1, min(nsplit - 1, s0));
        if (split < s0) { im = 0; nsplit = s0; }
        else { im = 1; split -= s0; nsplit -= s0; }
    }
    extern __shared__ int acc[]; // nb * AW votes (rows lo .. hi of this slab) + 64 spare words for lanes without an angle
    const int AW = 1 << aw_log2;
    int a0 = slab * AW;
    int na = min(AW, numangle - a0);
    const int lo = slab < VOTE_MAX_SLABS ? rng.lo[slab] : -((numrho - 1) / 2);
    const int nb = (slab < VOTE_MAX_SLABS ? rng.hi[slab] : numrho - 1 - (numrho - 1) / 2) - lo + 1;
    int n = cls ? (im ? nb_b : nb_e) : (im ? na_b : na_e);
    if ((size_t)n > list_cap) n = (int)list_cap;
    int per = ((n + nsplit - 1) / nsplit + 63) / 64 * 64;
    int begin = min(n, split * per), end = min(n, begin + per);
    const bool merge = balance || nsplit > 1; // pieces of a list are merged with atomic adds into a zeroed accumulator
    if (merge && begin >= end) return; // nothing to add
    const uint32_t *list = (im ? list1 : list0) + ((size_t)g * 2 + cls) * list_cap;
    int lane = threadIdx.x & 63;
    int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // the wave's first 64 list entries and the lane's trig values are on their way while the slab is being zeroed
    const int NT = (int)blockDim.x; // (the host picks the workgroup size: 1024 threads by default)
    const int nw_ = NT / 64, share_ = (end - begin + nw_ - 1) / nw_;
    const int wbeg_ = begin + wv * share_, wend_ = min(end, wbeg_ + share_);
    const uint32_t pv_first = (wbeg_ + lane < wend_) ? list[wbeg_ + lane] : 0u;
    float c_pre = 0.f, s_pre = 0.f;
    if ((lane & (AW - 1)) < na) { c_pre = tab[a0 + (lane & (AW - 1))]; s_pre = tab[numangle + a0 + (lane & (AW - 1))]; }
    for (int k = threadIdx.x; k < nb * AW + 64; k += NT) acc[k] = 0;
    __syncthreads();
    // lane = (chunk slot, angle): a slab of AW < 64 angles (large accumulators: the slab's rows x AW must fit in LDS) lets
    // every wave work on 64 / AW chunks side by side instead of idling the lanes beyond AW
    constexpr int SUBS = 64 >> aw_log2;
    const int ang = lane & (AW - 1), sub = lane >> aw_log2;
    bool act = ang < na;
    const float c = c_pre, s = s_pre;
    const int nw = NT / 64;
    // |r| <= (numrho-1)/2 by construction (numrho ~ 2(w+h)/rho, |j cos + i sin| < w+h), as in
    // OpenCV, which indexes its accumulator without a range check.
    // Vote address (bytes) = (rint(v) << (aw_log2+2)) + lanebase.  rint (round-half-even,
    // |v| < 2^22) uses the 1.5*2^23 trick: bits(v + 12582912.f) = 0x4B400000 + rint(v); the
    // constant is folded into the base (address arithmetic is mod 2^32).  Lanes without an angle
    // have c = s = 0, hence rint = 0, and their base points at the spare words: no select.
    // (an inactive lane votes bin rint(0) = 0: its base is shifted so that lands in the spare words)
    const unsigned cell = act ? (unsigned)((-lo) * AW + ang) : (unsigned)(nb * AW + lane);
    const unsigned lanebase = (cell << 2) - (0x4B400000u << (aw_log2 + 2));
    char *accb = (char *)acc;
    // A list entry is a chunk of horizontal neighbours (y, x0 .. x0 + len - 1), (len - 1) |c| < 1 for every angle of this slab (k_pixlist).
    // Along a chunk the vote value w(x) = fl(fl(x c) + y s) is monotone in x and moves by less than one bin
    // (|c| = |cos| / rho per pixel), so the chunk's pixels fall into at most two adjacent bins: the first t
    // into bin(x0), the rest into bin(x0 + len - 1).  Per angle:
    //   both ends in one bin            -> one LDS add of len;
    //   otherwise t is estimated from the line equation, t = ceil((r0 +- 1/2 - w0) / c), and CHECKED with the
    //   exact per-pixel arithmetic at x0 + t - 1 and x0 + t (bin(x0) and bin(x_end) respectively): monotone
    //   + both checks hold <=> t is exact.  If any lane's check fails (ties, rounding of the estimate) the
    //   wave walks the chunk pixel by pixel.  Either way every pixel's bin is the one OpenCV computes.
    // ~40 vector instructions and 2 LDS adds per chunk and wave instead of 6 and 1 per pixel.
    const float inv_c = 1.0f / c;                 // +-inf for lanes without an angle (c = 0): harmless, see below
    const float half_s = c < 0.f ? -0.5f : 0.5f;
    const int sh = aw_log2 + 2;
#define LFD_BIN(FX, YS) __builtin_bit_cast(unsigned, __fadd_rn(__fadd_rn(__fmul_rn((FX), c), (YS)), 12582912.0f))
    // every wave takes one contiguous share of the piece (a strided walk in batches of 64 left most waves idle in the last
    // round: 1 250 chunks over 16 waves were two batches for four waves and one for the other twelve)
    const int share = (end - begin + nw - 1) / nw;
    const int wbeg = begin + wv * share, wend = min(end, wbeg + share);
    for (int base = wbeg; base < wend; base += 64) {
        int m = min(64, wend - base);
        uint32_t pv = base == wbeg ? pv_first : ((lane < m) ? list[base + lane] : 0u);
        // every lane converts its own entry once; the wave then walks the entries with v_readlane (one chunk per step,
        // SUBS == 1) or fetches its slot's entry with ds_bpermute (SUBS chunks per step)
        const int lenv = (int)(pv >> 26);                       // len - 1
        const float fx0v = (float)(pv & 0x1fffu), fyv = (float)((pv >> 13) & 0x1fffu);
        const float fxev = fx0v + (float)lenv;
        for (int k = 0; k < m; k += SUBS) {
            float fx0, fxe, fy;
            int L;
            bool on = true; // this lane's slot holds a chunk
            if constexpr (SUBS == 1) {
                fx0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fx0v), k));
                fxe = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fxev), k));
                fy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fyv), k));
                L = __builtin_amdgcn_readlane(lenv, k) + 1;
            } else {
                const int src = k + sub;
                on = src < m;
                const uint32_t pe = (uint32_t)__builtin_amdgcn_ds_bpermute((src & 63) << 2, (int)pv);
                L = (int)(pe >> 26) + 1;
                fx0 = (float)(pe & 0x1fffu);
                fy = (float)((pe >> 13) & 0x1fffu);
                fxe = fx0 + (float)(L - 1);
            }
            const float ys = __fmul_rn(fy, s);
            const float w0 = __fadd_rn(__fmul_rn(fx0, c), ys);
            const float v0 = __fadd_rn(w0, 12582912.0f);
            const unsigned b0 = __builtin_bit_cast(unsigned, v0), be = LFD_BIN(fxe, ys);
            if (__ballot(on && b0 != be) == 0ull) { // every chunk of this step votes for one bin at every angle of this wave
                if (on) atomicAdd((int *)(accb + ((b0 << sh) + lanebase)), L);
                continue;
            }
            // boundary estimate (lanes whose ends agree get some t in [1, L-1]: both parts land in the same bin)
            float tf = __fmul_rn(__fsub_rn(__fadd_rn(__fsub_rn(v0, 12582912.0f), half_s), w0), inv_c);
            tf = fminf(fmaxf(tf, 1.0f), (float)(L - 1)); // also takes care of inf / NaN (lanes without an angle)
            const int t = (int)ceilf(tf);
            const float fxb = __fadd_rn(fx0, (float)t), fxa = __fsub_rn(fxb, 1.0f);
            const unsigned ba = LFD_BIN(fxa, ys), bb = LFD_BIN(fxb, ys);
            const bool ok = !on || b0 == be || ((ba == b0) && (bb == be)); // (both ends in one bin: any split adds up to L there)
            if (__ballot(!ok) == 0ull) {
                if (on) {
                    atomicAdd((int *)(accb + ((b0 << sh) + lanebase)), t);
                    atomicAdd((int *)(accb + ((be << sh) + lanebase)), L - t);
                }
                continue;
            }
            // One pixel off is common where bin borders sit on whole pixels (theta = 0: x cos / rho crosses
            // r + 1/2 at x = rho r + rho / 2, and whether float arithmetic lands above or below is a coin
            // toss): look one pixel further on the side that disagreed.
            const bool high = (ba == be); // the estimate overshot: x0 + t - 1 is already in the second bin
            const unsigned bc = LFD_BIN(high ? __fsub_rn(fxa, 1.0f) : __fadd_rn(fxb, 1.0f), ys);
            const int t2 = ok ? t : (high ? t - 1 : t + 1);
            const bool ok2 = ok || (high ? (bc == b0) : (ba == b0 && bb == b0 && bc == be));
            if (__ballot(!ok2) == 0ull) {
                if (on) {
                    atomicAdd((int *)(accb + ((b0 << sh) + lanebase)), t2);
                    atomicAdd((int *)(accb + ((be << sh) + lanebase)), L - t2);
                }
                continue;
            }
            int Lmax = L; // rare: exact walk (the longest chunk of the step sets the trip count)
            if constexpr (SUBS > 1) {
                for (int off = 32; off > 0; off >>= 1) Lmax = max(Lmax, __shfl_xor(Lmax, off));
            }
            for (int j = 0; j < Lmax; j++)
                if (on && j < L) atomicAdd((int *)(accb + ((LFD_BIN(__fadd_rn(fx0, (float)j), ys) << sh) + lanebase)), 1);
        }
    }
#undef LFD_BIN
    __syncthreads();
    int *ag = accum + ((size_t)g * 2 + im) * acc_cap;
    const int ts = numangle + 2; // transposed row length
    const int r0 = (numrho - 1) / 2 + lo; // accumulator row of the slab's first bin
    if (merge) {
        for (int k = threadIdx.x; k < nb * AW; k += NT) {
            int rr = r0 + (k >> aw_log2), al = k & (AW - 1);
            int v = acc[k];
            if (v && al < na) atomicAdd(&ag[(size_t)(rr + 1) * ts + a0 + al + 1], v);
        }
        return;
    }
    for (int k = threadIdx.x; k < numrho * AW; k += NT) { // rows outside lo .. hi are zero
        int rr = k >> aw_log2, al = k & (AW - 1);
        int rl = rr - r0;
        if (al < na) ag[(size_t)(rr + 1) * ts + a0 + al + 1] = (rl >= 0 && rl < nb) ? acc[rl * AW + al] : 0;
    }
    // guard cells: bins -1 and numrho for this slab's angles; angles -1 and numangle for all bins
    for (int k = threadIdx.x; k < na; k += NT) {
        ag[a0 + k + 1] = 0;
        ag[(size_t)(numrho + 1) * ts + a0 + k + 1] = 0;
    }
    if (slab == 0)
        for (int k = threadIdx.x; k < numrho + 2; k += NT) { ag[(size_t)k * ts] = 0; ag[(size_t)k * ts + ts - 1] = 0; }
}

// findLocalMaximums on the transposed accumulator: key = votes << 32 | (0x7fffffff - base) with
// OpenCV's base = (n+1)*(numrho+2) + r+1, so that a descending sort of the keys is OpenCV's
// hough_cmp_gt order (votes desc, base asc).  Enumeration order is irrelevant.
__global__ void __launch_bounds__(256)
k_hough_peaks(const int *accum, u64 *peaks, int *counters, int numangle, int numrho, int threshold,
              size_t acc_cap, size_t peak_cap, const int *active, int need_detect) {
    int g = blockIdx.z, im = blockIdx.y;
    if (slot_off(active, counters, g)) return;
    int *cnt = counters + g * C_COUNT;
    if (need_detect && !cnt[C_DETECT]) return;
    const int *ag = accum + ((size_t)g * 2 + im) * acc_cap;
    u64 *pg = peaks + ((size_t)g * 2 + im) * peak_cap;
    const int ts = numangle + 2;
    // Peaks are collected in LDS and handed to the frame's list with ONE global atomic per flush: with threshold 1 an
    // accumulator has thousands of local maxima, and one atomicAdd each on the same counter serialises (~100 ns apiece).
    constexpr int PK_BUF = 4096; // (a block of rows adds at most 8 x 256 entries: flushed while that many are free)
    __shared__ u64 buf[PK_BUF];
    __shared__ int nbuf, gbase;
    if (threadIdx.x == 0) nbuf = 0;
    __syncthreads();
    auto flush = [&]() { // (all threads; nbuf is stable: callers sit between two barriers)
        const int nb = nbuf;
        if (threadIdx.x == 0 && nb) gbase = atomicAdd(&cnt[im ? C_NPEAK_BOX : C_NPEAK_EQU], nb);
        __syncthreads();
        for (int i = threadIdx.x; i < nb; i += (int)blockDim.x) {
            if ((size_t)(gbase + i) < peak_cap) pg[gbase + i] = buf[i];
            else cnt[C_OVERFLOW] = 1; // a truncated peak list could lose a top line: run the frame again (spill workspace)
        }
        __syncthreads();
        if (threadIdx.x == 0) nbuf = 0;
        __syncthreads();
    };
    // a thread owns an angle column and walks a band of bins, RB rows of loads in flight; the bins above and below come from
    // registers, the neighbouring angles from the neighbouring lanes (the first and last lane of a wave load theirs)
    constexpr int RB = 8;
    const int per = ((numrho + (int)gridDim.x - 1) / (int)gridDim.x + RB - 1) / RB * RB;
    const int r0 = blockIdx.x * per, r1 = min(numrho, r0 + per);
    const int lane = threadIdx.x & 63, nth = (int)blockDim.x;
    int blk = 0;
    for (int n0 = 0; n0 < numangle; n0 += nth) {
        const int n = n0 + threadIdx.x;
        const bool on = n < numangle;
        const int *col = ag + (on ? n : 0) + 1;
        for (int rb = r0; rb < r1; rb += RB, blk++) {
            int v[RB + 2];
#pragma unroll
            for (int j = 0; j < RB + 2; j++) { // guarded rows rb .. rb + RB + 1 = bins rb - 1 .. rb + RB
                const int rr = rb + j;
                v[j] = (on && rr <= numrho + 1) ? col[(size_t)rr * ts] : 0;
            }
#pragma unroll
            for (int j = 0; j < RB; j++) {
                const int r = rb + j, c = v[j + 1];
                int left = __shfl_up(c, 1), right = __shfl_down(c, 1);
                const bool cand = on && r < r1 && c > threshold && c > v[j] && c >= v[j + 2];
                if (cand) {
                    const int t = (r + 1) * ts + n + 1;
                    if (lane == 0) left = ag[t - 1];
                    if (lane == 63 || n == numangle - 1) right = ag[t + 1];
                    // accum[base-1], [base+1]: r -/+ 1 (above);  accum[base -/+ (numrho+2)]: angle -/+ 1
                    if (c > left && c >= right) {
                        const int base = (n + 1) * (numrho + 2) + r + 1;
                        buf[atomicAdd(&nbuf, 1)] = ((u64)(uint32_t)c << 32) | (u64)(uint32_t)(0x7fffffff - base);
                    }
                }
            }
            __syncthreads();
            if (nbuf > PK_BUF - RB * 256) flush();
            else __syncthreads();
        }
    }
    __syncthreads();
    flush();
}

// transposed accumulator -> OpenCV layout (stand-alone lfdmi_hough_accum only)
__global__ void __launch_bounds__(256)
k_accum_untranspose(const int *accum, int *out, int numangle, int numrho, size_t acc_cap) {
    int g = blockIdx.y;
    const int *ag = accum + (size_t)g * 2 * acc_cap;
    int *og = out + (size_t)g * (numangle + 2) * (numrho + 2);
    int total = (numangle + 2) * (numrho + 2);
    for (int k = blockIdx.x * 256 + threadIdx.x; k < total; k += gridDim.x * 256) {
        int n1 = k / (numrho + 2), r1 = k - n1 * (numrho + 2);
        og[k] = ag[(size_t)r1 * (numangle + 2) + n1];
    }
}

__device__ __forceinline__ void key_to_line(u64 key, int numrho, float rho, float theta, float *orho,
                                            float *otheta) {
    int idx = 0x7fffffff - (int)(uint32_t)(key & 0xffffffffu);
    float scale = __fdiv_rn(1.f, (float)(numrho + 2));
    int n = (int)floorf(__fmul_rn((float)idx, scale)) - 1;
    int r = idx - (n + 1) * (numrho + 2) - 1;
    *orho = __fmul_rn(__fsub_rn((float)r, __fmul_rn((float)(numrho - 1), 0.5f)), rho);
    *otheta = __fadd_rn(0.f, __fmul_rn((float)n, theta));
}

// top-K selection (K small): K rounds of "largest key below the previous one".
// One workgroup per (image, slot).  lines: [G][2][K][2] floats.
__global__ void __launch_bounds__(256)
k_hough_topk(const u64 *peaks, const int *counters, float *lines, int K, int numrho, float rho,
             float theta, size_t peak_cap, const int *active, int need_detect) {
    int g = blockIdx.y, im = blockIdx.x;
    if (slot_off(active, counters, g)) return;
    const int *cnt = counters + g * C_COUNT;
    if (need_detect && !cnt[C_DETECT]) return;
    const u64 *pg = peaks + ((size_t)g * 2 + im) * peak_cap;
    int n = cnt[im ? C_NPEAK_BOX : C_NPEAK_EQU];
    if ((size_t)n > peak_cap) n = (int)peak_cap;
    __shared__ u64 red[4];
    __shared__ u64 prev_s;
    u64 prev = ~0ull;
    float *lg = lines + ((size_t)g * 2 + im) * K * 2;
    for (int round = 0; round < K; round++) {
        u64 best = 0;
        for (int k = threadIdx.x; k < n; k += 256) {
            u64 v = pg[k];
            if (v < prev && v > best) best = v;
        }
        for (int off = 32; off > 0; off >>= 1) {
            u64 o = __shfl_down(best, off);
            if (o > best) best = o;
        }
        if (lfd_lane() == 0) red[threadIdx.x >> 6] = best;
        __syncthreads();
        if (threadIdx.x == 0) {
            u64 b = red[0];
            for (int k = 1; k < 4; k++) if (red[k] > b) b = red[k];
            prev_s = b;
            float ro = 0.f, th = 0.f;
            if (b) key_to_line(b, numrho, rho, theta, &ro, &th);
            lg[2 * round] = ro;
            lg[2 * round + 1] = th;
        }
        __syncthreads();
        prev = prev_s;
        if (!prev) { // fewer than K lines: zero-fill the rest
            if (threadIdx.x == 0)
                for (int r2 = round + 1; r2 < K; r2++) { lg[2 * r2] = 0.f; lg[2 * r2 + 1] = 0.f; }
            break;
        }
    }
}

// full ordering for the standalone HoughLines entry point: single-workgroup bitonic sort
// (descending) of the peak keys in global memory, then conversion of the first max_lines.
__global__ void __launch_bounds__(1024)
k_hough_sort(u64 *peaks, const int *counters, float *lines, int max_lines, int numrho, float rho,
             float theta, size_t peak_cap) {
    int g = blockIdx.y, im = blockIdx.x;
    const int *cnt = counters + g * C_COUNT;
    if (cnt[C_OVERFLOW]) return; // truncated lists: the host runs the image again through the worst-case workspace
    u64 *pg = peaks + ((size_t)g * 2 + im) * peak_cap;
    int n = cnt[im ? C_NPEAK_BOX : C_NPEAK_EQU];
    if ((size_t)n > peak_cap) n = (int)peak_cap;
    int np2 = 1;
    while (np2 < n) np2 <<= 1;
    for (int k = n + threadIdx.x; k < np2; k += 1024) pg[k] = 0;
    __syncthreads();
    for (int size = 2; size <= np2; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = threadIdx.x; t < (np2 >> 1); t += 1024) {
                int lo = 2 * t - (t & (stride - 1)); // index with bit `stride` clear
                int hi = lo + stride;
                bool desc = ((lo & size) == 0);
                u64 a = pg[lo], b = pg[hi];
                if (desc ? (a < b) : (a > b)) { pg[lo] = b; pg[hi] = a; }
            }
            __syncthreads();
        }
    int m = min(n, max_lines);
    float *lg = lines + (size_t)g * max_lines * 2;
    for (int k = threadIdx.x; k < m; k += 1024) key_to_line(pg[k], numrho, rho, theta, &lg[2 * k], &lg[2 * k + 1]);
}

// ---- numpy add.reduce on a short float64 vector: first + pairwise(rest) -------------------
__device__ double np_pairwise_dev(const double *a, int n) {
    if (n < 8) {
        double res = 0.;
        for (int i = 0; i < n; i++) res = __dadd_rn(res, a[i]);
        return res;
    }
    // n <= LFDMI_MAX_SET_LINES - 1 < 128: the 8-accumulator block of numpy's pairwise sum
    double r[8], res;
    int i;
    for (int j = 0; j < 8; j++) r[j] = a[j];
    for (i = 8; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; j++) r[j] = __dadd_rn(r[j], a[i + j]);
    res = __dadd_rn(__dadd_rn(__dadd_rn(r[0], r[1]), __dadd_rn(r[2], r[3])),
                    __dadd_rn(__dadd_rn(r[4], r[5]), __dadd_rn(r[6], r[7])));
    for (; i < n; i++) res = __dadd_rn(res, a[i]);
    return res;
}
__device__ double np_mean_dev(const double *a, int n) {
    return __ddiv_rn(__dadd_rn(a[0], np_pairwise_dev(a + 1, n - 1)), (double)n);
}

struct TailParams {
    int navg;
    double dro, thetaTresh, lineSetTresh;
    int which; // 1 bright, 2 dim
};

// One thread per slot: check_theta on the first navg lines of both sets, result record,
// and the "dim pass needed" flag (detecttrails.py:125-131).
__global__ void k_finalize(const float *lines, const int *counters, lfdmi_result *res, int *need_dim,
                           int *pass_flags, const int *active, TailParams tp, int G) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    if (active && !active[g]) { if (need_dim) need_dim[g] = 0; return; }
    const int *cnt = counters + g * C_COUNT;
    lfdmi_result r = res[g];
    r.detection = cnt[C_DETECT];
    r.rejected_by_theta = 0;
    r.n_lines_equ = r.n_lines_box = 0;
    if (cnt[C_OVERFLOW]) {
        // a table was too small for this frame: the host enlarges the tables (or runs the frame alone in the worst-case
        // workspace); what the frame asked for travels in the fields an overflowed record has no use for
        r.status = LFDMI_ERR_CAPACITY;
        r.x1 = max(cnt[C_NRUNF], cnt[C_NRUNB]);                       // runs
        r.y1 = cnt[C_NKEYS];                                          // contour keys (clamped at the capacity)
        r.x2 = cnt[C_NSLOTS];                                         // row slots
        r.y2 = max(max(cnt[C_NPIX_EQU], cnt[C_NPIX_BOX]), max(cnt[C_NPIXB_EQU], cnt[C_NPIXB_BOX])); // Hough list entries
        r.n_lines_equ = max(cnt[C_NPEAK_EQU], cnt[C_NPEAK_BOX]);      // accumulator peaks
    }
    if (r.detection && r.status == 0) {
        int n1 = cnt[C_NPEAK_EQU], n2 = cnt[C_NPEAK_BOX];
        r.n_lines_equ = n1; r.n_lines_box = n2;
        if (n1 == 0 || n2 == 0) {
            r.status = LFDMI_ERR_NOLINES;
        } else {
            int K = tp.navg;
            const float *l1 = lines + ((size_t)g * 2 + 0) * K * 2, *l2 = lines + ((size_t)g * 2 + 1) * K * 2;
            double ro1[LFDMI_MAX_SET_LINES], ro2[LFDMI_MAX_SET_LINES], t1[LFDMI_MAX_SET_LINES],
                t2[LFDMI_MAX_SET_LINES];
            for (int i = 0; i < K; i++) {
                ro1[i] = ro2[i] = t1[i] = t2[i] = 0.;
                if (i >= n1) continue;
                ro1[i] = l1[2 * i];
                if (i >= n2) continue;
                ro2[i] = l2[2 * i];
                t1[i] = l1[2 * i + 1];
                t2[i] = l2[2 * i + 1];
            }
            bool rej = fabs(__dsub_rn(np_mean_dev(ro1, K), np_mean_dev(ro2, K))) > tp.dro;
            if (!rej) {
                double mx = t1[0], mn = t1[0];
                for (int i = 1; i < K; i++) { mx = fmax(mx, t1[i]); mn = fmin(mn, t1[i]); }
                rej = fabs(__dsub_rn(mx, mn)) > tp.thetaTresh;
            }
            if (!rej) {
                double mx = t2[0], mn = t2[0];
                for (int i = 1; i < K; i++) { mx = fmax(mx, t2[i]); mn = fmin(mn, t2[i]); }
                rej = fabs(__dsub_rn(mx, mn)) > tp.thetaTresh;
            }
            if (!rej) {
                for (int i = 0; i < K; i++) ro1[i] = fabs(__dsub_rn(t1[i], t2[i]));
                rej = np_mean_dev(ro1, K) > tp.lineSetTresh;
            }
            if (rej) r.rejected_by_theta = 1;
            else { r.found = tp.which; r.rho = l1[0]; r.theta = l1[1]; }
        }
    }
    res[g] = r;
    pass_flags[g] |= (tp.which == 1) ? (r.detection ? 1 : 0) : (2 | (r.detection ? 4 : 0));
    if (need_dim) need_dim[g] = (r.found == 0 && r.status == 0) ? 1 : 0;
}

// multi-scale Hough: the list / peak counters of the previous rho are cleared before the next one is evaluated
// (C_OVERFLOW stays: a frame that overflowed at any scale is run again as a whole)
__global__ void k_hough_reset(int *counters, int G) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    int *cnt = counters + g * C_COUNT;
    cnt[C_NPIX_EQU] = 0; cnt[C_NPIX_BOX] = 0; cnt[C_NPIXB_EQU] = 0; cnt[C_NPIXB_BOX] = 0; cnt[C_NPEAK_EQU] = 0; cnt[C_NPEAK_BOX] = 0;
    cnt[C_NNZ_EQU] = 0; cnt[C_NNZ_BOX] = 0;
}

__global__ void k_init_results(lfdmi_result *res, int *pass_flags, int G) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    pass_flags[g] = 0;
    lfdmi_result r;
    r.status = 0; r.found = 0; r.rho = 0.f; r.theta = 0.f;
    r.x1 = r.y1 = r.x2 = r.y2 = 0;
    r.n_lines_equ = r.n_lines_box = 0; r.detection = 0; r.rejected_by_theta = 0;
    res[g] = r;
}
