// k_frame.h -- per-frame connectivity with the run tables in LDS.
//
// The run kernels of k_ccl.h spread one frame's few thousand active words over 48 workgroups
// and chase union-find links through HBM: every step of a `find` is a ~2 us round trip and the
// whole chain (init -> merge -> flatten -> edges) is five launches of mostly waiting waves.  A
// sky frame has only a few thousand runs, so here ONE workgroup owns a frame and keeps the
// label table in LDS (a `find` step costs an LDS access), walking the same work list through
// the phases with workgroup barriers in between.  Results are identical: the union-find links
// always point to the smaller id, so every component's root is its raster-first run either way.
// A frame with more runs than the LDS table holds is left to the multi-workgroup kernels of
// k_ccl.h: each k_frame_* kernel writes a per-frame `fallback` flag those are launched with as
// their `active` mask (they exit at once when the flag is clear), and raises PASS_FLAG_GENERAL in
// the frame's pass flags.  The host launches the general kernels only once a context has seen such
// a frame (22 empty launches per pass otherwise): a chunk that raises the flag while they were not
// launched is simply run again with them (lfdmi.hip: general_on).
//
// Every phase walks the work list through frame_pipeline(): the global loads of an item (its
// words, their scan entries) do not depend on any table, so the loads of the NEXT item and the
// list entry after that are issued before the current item is processed -- the ~2 us HBM round
// trips overlap with the LDS work instead of adding up.
#pragma once
#include "k_ccl.h"
#include "k_rect.h"

// perm[0 .. na) = the slots with active[slot] != 0 in ascending order, perm[na .. n) = the others: kernels whose workgroup ->
// frame mapping decides which XCD a frame works on (one workgroup per frame; the tile kernel's frame = f(block % 8)) index the
// frames through it, so a pass that works on an arbitrary subset of the slots still loads the eight XCDs evenly.
__global__ void __launch_bounds__(64) k_active_perm(const int *active, int n, int *perm) {
    const int lane = threadIdx.x;
    int base = 0;
    for (int pass = 0; pass < 2; pass++)
        for (int i0 = 0; i0 < n; i0 += 64) {
            const int i = i0 + lane;
            const bool on = i < n && ((active[i] != 0) == (pass == 0));
            const u64 bal = __ballot(on);
            if (on) perm[base + __popcll(bal & ((1ull << lane) - 1ull))] = i;
            base += __popcll(bal);
        }
}

// The same for the tile kernel, whose frames are pinned to an XCD each (frame i of the list -> XCD i % 8): active frames sorted
// by their number of active tiles, most first, and dealt to the XCDs in snake order (0..7, 7..0, ...), so that the eight XCDs
// get nearly equal sums of tiles instead of whatever 32 consecutive frames happen to hold.  n <= 1024.
__global__ void __launch_bounds__(1024) k_tile_perm(const int *active, const int *counters, int n, int *perm) {
    __shared__ int key[1024];
    const int i = threadIdx.x;
    if (i < n) key[i] = (active && !active[i]) ? -1 : counters[i * C_COUNT + C_NTILES];
    __syncthreads();
    if (i >= n) return;
    const int k = key[i];
    int rank = 0;
    for (int j = 0; j < n; j++) {
        const int kj = key[j];
        rank += (kj > k || (kj == k && j < i)) ? 1 : 0;
    }
    const int round = rank >> 3, pos = rank & 7;
    int c = (round << 3) + ((round & 1) ? 7 - pos : pos);
    if (c >= n) c = rank; // (the last, incomplete round keeps its order: positions beyond n do not exist)
    perm[c] = i;
}

#define PASS_FLAG_GENERAL 256 // pass_flags bit: this frame needs the general (multi-workgroup) run kernels
#define FRAME_THREADS 1024
#define FRAME_HOLECAP 1024 // hash slots for the holes of a frame (more holes: looked up in the global tables)
#define FRAME_RUNCAP 32768 // runs per frame the LDS label table holds (128 KB); busier frames take the k_ccl.h kernels

__device__ __forceinline__ int lds_find(const int *L, int x) {
    int p;
    while ((p = L[x]) != x) x = p;
    return x;
}

// find with path halving: a non-root entry may be re-pointed to any ancestor at any time (links
// only ever go to smaller ids, and a linked run never becomes a root again)
__device__ __forceinline__ int lds_find_halve(int *L, int x) {
    int p;
    while ((p = L[x]) != x) {
        int gp = L[p];
        if (gp != p) L[x] = gp;
        x = gp;
    }
    return x;
}

__device__ __forceinline__ void lds_union(int *L, int a, int b) {
    for (;;) {
        a = lds_find_halve(L, a);
        b = lds_find_halve(L, b);
        if (a == b) return;
        if (a < b) { int t = a; a = b; b = t; }
        int old = atomicMin(&L[a], b);
        if (old == a) return;
        a = old;
    }
}

// (Round 4, measured and dropped: hooking without the two finds -- one atomicMin per contact, then carrying on with the larger
// id's former parent -- walks the whole chain of a tall component on every later contact: merge 72 -> 172 us (median, a frame's
// maximum 1 ms); and flattening by pointer jumping after the merge -- log2(rows) rounds of two LDS reads per run -- costs more
// rounds than the walks it saves once the finds halve their paths: 4.44 vs 4.43 ms per step.  profiles/README.md.)
// (Taking the items four at a time -- all of a block's loads issued back to back, the next block's list entries fetched
// meanwhile -- measured 5-7 % SLOWER for both per-frame kernels: their phases are bound by the LDS label chains and the
// memory-side atomics of the slot updates, not by the exposed HBM round trips, and the extra live registers cost more.)
// Which list positions a thread walks: start, start + step, ... below limit.
//   strided: thread t takes t, t + 1024, ... (every phase touches all rows of the frame at once);
//   blocked: wave v takes a contiguous 1/16 of the (raster-ordered) list, 64 items per step: a wave works its band of rows
//            top to bottom, so by the time a run is joined to the row above, that row's runs have mostly found their roots
//            -- the union-find chains the merge phases chase stay short.
struct ItemMap { int start, step, limit; };
__device__ __forceinline__ ItemMap item_map(int nwork, int blocked) {
    ItemMap m;
    if (blocked) {
        const int per = ((nwork + FRAME_THREADS - 1) / FRAME_THREADS) * 64, wv = threadIdx.x >> 6;
        m.start = wv * per + (threadIdx.x & 63); m.step = 64; m.limit = min(nwork, (wv + 1) * per);
    } else { m.start = threadIdx.x; m.step = FRAME_THREADS; m.limit = nwork; }
    return m;
}

template <class Item, class LoadF, class ProcF>
__device__ __forceinline__ void frame_pipeline(const int *wl, ItemMap m, int k0, LoadF load, ProcF proc) {
    int it = m.start + k0 * m.step;
    int i1 = it < m.limit ? wl[it] : -1;
    int i2 = it + m.step < m.limit ? wl[it + m.step] : -1;
    Item cur;
    if (i1 >= 0) cur = load(i1);
    while (i1 >= 0) {
        int i3 = it + 2 * m.step < m.limit ? wl[it + 2 * m.step] : -1;
        Item nxt;
        if (i2 >= 0) nxt = load(i2);
        proc(cur);
        cur = nxt;
        i1 = i2;
        i2 = i3;
        it += m.step;
    }
}
template <class Item, class LoadF, class ProcF>
__device__ __forceinline__ void frame_pipeline(const int *wl, int nwork, LoadF load, ProcF proc) {
    ItemMap m = {(int)threadIdx.x, FRAME_THREADS, nwork};
    frame_pipeline<Item>(wl, m, 0, load, proc);
}

// bits 0 .. b
__device__ __forceinline__ u64 upto_bit(int b) { return (b == 63) ? ~0ull : ((2ull << b) - 1ull); }

typedef u64 u64x2a8 __attribute__((ext_vector_type(2), aligned(8))); // two neighbouring words of a bit row: one 16-byte load
// base[i - 1], base[i] with one load (prev = `none` when the word has no left neighbour in its row)
__device__ __forceinline__ void word_pair(const u64 *base, int i, bool with_prev, u64 none, u64 &prev, u64 &cur) {
    if (with_prev) {
        const u64x2a8 v = *(const u64x2a8 *)(base + i - 1);
        prev = v.x; cur = v.y;
    } else { prev = none; cur = base[i]; }
}

struct FgMergeItem { int idx, id0, idu; u64 c, cp, u, up, un; };
struct FgWordItem { int idx, id0; u64 c, cp, m; };

// Hysteresis of one frame: 8-connected components of the candidate runs, strong flags, edge
// bit rows; Lf / YMf / FLf / ROWf written for the contour kernels.  Replaces k_runs_init(fg),
// k_runs_merge8, k_runs_flatten(fg) and k_edge_from_cand.
__global__ void __launch_bounds__(FRAME_THREADS)
k_frame_fg(const u64 *cand, const u64 *strong, const int *scanf, const int *wl_fg, int *counters, int *Lf, int *YMf,
           int *FLf, int *ROWf, u64 *edge, int h, int w, int run_cap, int lds_cap, const int *active, int *fallback,
           int *pass_flags, int lds_ints, const int *perm, long long *prof,
           int4 *keys, int *bigkeys, int *medkeys, int2 *rowext, int key_cap, int slot_cap, int *fg_keys, int blocked) {
    // lds_ints: ints of dynamic LDS this launch allocated.  keys != nullptr: the contour stage follows (k_frame_contours): the
    // OUTER-border keys of the edge components, their row slots and the per-row extremes are made here, where the labels of the
    // candidate runs already sit in LDS (round 4; until then k_frame_contours re-read them from memory: three of its phases and
    // a gather per edge stretch), and fg_keys[frame] = 1 tells k_frame_contours so.
    const int g = perm ? perm[blockIdx.x] : (int)blockIdx.x; // (active frames first: one workgroup per frame, XCD = workgroup % 8)
    const long long t0 = prof ? wall_clock64() : 0;
    int pk = 8;
#define FG_PROF() do { if (prof) { __syncthreads(); if (threadIdx.x == 0) prof[g * 16 + (pk++)] = wall_clock64() - t0; } } while (0)
    if (slot_off(active, counters, g)) {
        if (threadIdx.x == 0) { fallback[g] = 0; if (fg_keys) fg_keys[g] = 0; }
        return;
    }
    const int wq = LFD_WQ(w);
    const size_t fo = (size_t)g * h * wq, ro = (size_t)g * run_cap;
    const int nwork = counters[g * C_COUNT + C_NFGW], nrun = counters[g * C_COUNT + C_NRUNF];
    // LDS: labels L[n32] | strong-root bits | "a run below" bits | X: the rest (last rows per root, then the row slots)
    const int n32 = (nrun + 31) & ~31;
    const int sizeX = lds_ints - n32 - 2 * (n32 / 32);
    const bool fits = nrun <= lds_cap && sizeX >= 0; // lds_cap <= FRAME_RUNCAP (k_scan_bases flags nrun > run_cap as an overflow)
    const bool keys_path = keys != nullptr && fits && sizeX >= n32; // (room for a last-row entry per run)
    if (threadIdx.x == 0) {
        fallback[g] = fits ? 0 : 1;
        if (fg_keys) fg_keys[g] = keys_path ? 1 : 0;
        if (!fits) atomicOr(&pass_flags[g], PASS_FLAG_GENERAL); // tells the host the general kernels are needed
    }
    if (!fits) return;
    extern __shared__ int sm_frame[];
    int *L = sm_frame;
    unsigned *FL = (unsigned *)(sm_frame + n32);         // root holds a strong pixel
    unsigned *HB = FL + n32 / 32;                        // run touches a run of the next row
    int *X = (int *)(HB + n32 / 32);
    int *YM = X;                                         // keys_path: last row of every root's component
    __shared__ int c_slots, c_keys, c_big, c_med, c_ovf;
    if (threadIdx.x == 0) { c_slots = 0; c_keys = 0; c_big = 0; c_med = 0; c_ovf = 0; }
    int *YMg = YMf + ro, *ROWg = ROWf + ro;
    const u64 *fb = cand + fo, *mb = strong + fo;
    const int *sf = scanf + fo, *wl = wl_fg + fo;
    for (int i = threadIdx.x; i < (nrun + 31) / 32; i += FRAME_THREADS) { FL[i] = 0u; HB[i] = 0u; }
    for (int i = threadIdx.x; i < nrun; i += FRAME_THREADS) L[i] = i; // every run its own root
    __syncthreads();
    constexpr int FG_KC = 8;
    const ItemMap im = item_map(nwork, blocked & 1);
    const bool more = nwork > FG_KC * FRAME_THREADS; // (items beyond the register copies: gathered again in every phase)
    u64 kc_c[FG_KC];
    int kc_i[FG_KC], kc_x[FG_KC]; // id0 | (last pixel of the word to the left) << 31, idx
    // ---- row of every run; 8-connectivity between rows y and y-1
    auto merge_load = [&](int idx) {
            FgMergeItem t;
            int y = idx / wq, q = idx - y * wq;
            t.idx = idx;
            word_pair(fb, idx, q > 0, 0ull, t.cp, t.c);
            t.id0 = sf[idx];
            t.u = 0; t.up = 0; t.un = 0; t.idu = 0;
            if (y > 0) {
                word_pair(fb, idx - wq, q > 0, 0ull, t.up, t.u);
                t.un = q + 1 < wq ? fb[idx - wq + 1] : 0ull;
                t.idu = sf[idx - wq];
            }
            return t;
    };
    auto merge_proc = [&](const FgMergeItem &t) {
            int y = t.idx / wq, q = t.idx - y * wq;
            u64 vmask = valid_mask(q, w);
            u64 c = t.c & vmask;
            u64 s = c & ~((c << 1) | (t.cp >> 63)); // run starts of this word
            for (int k = 0, n = __popcll(s); k < n; k++) {
                if (keys_path) YM[t.id0 + k] = y; // (row and last row of the roots go to memory when their keys are made)
                else { ROWg[t.id0 + k] = y; YMg[t.id0 + k] = y; }
            }
            if (y == 0 || !c) return;
            u64 u = t.u & vmask;
            u64 su = u & ~((u << 1) | (t.up >> 63)); // run starts of the word above
            u64 uL = (u << 1) | (t.up >> 63);         // bit x set <=> up[x-1]
            u64 uR = (u >> 1) | (t.un << 63);         // bit x set <=> up[x+1]
            u64 v0 = c & u;
            v0 &= ~(v0 << 1); // first column of every vertical-contact stretch
            u64 vm = c & uL & ~u, vp = c & uR & ~u; // diagonal contacts not implied by a vertical one
            // run holding bit b of this word / of the word above (the bit is set)
            auto own = [&](int b) { return t.id0 + __popcll(s & upto_bit(b)) - 1; };
            auto above = [&](int b) { return t.idu + __popcll(su & upto_bit(b)) - 1; };
            // every touching (upper run, lower run) pair shows up in exactly these contacts, so the
            // upper run of a contact is never in its component's last row
            auto join = [&](int lo, int hi) {
                atomicOr(&HB[hi >> 5], 1u << (hi & 31));
                lds_union(L, lo, hi);
            };
            while (v0) {
                int b = __ffsll((long long)v0) - 1;
                v0 &= v0 - 1;
                join(own(b), above(b));
            }
            while (vm) {
                int b = __ffsll((long long)vm) - 1;
                vm &= vm - 1;
                // bit b-1 of the word above; for b == 0 the last pixel of the previous word: the run
                // started last before this word
                join(own(b), b ? above(b - 1) : t.idu - 1);
            }
            while (vp) {
                int b = __ffsll((long long)vp) - 1;
                vp &= vp - 1;
                // bit b+1 of the word above; for b == 63 the first pixel of the next word, which starts a
                // run there (bit 63 above is clear): the first run after those starting in this word
                join(own(b), b < 63 ? above(b + 1) : t.idu + __popcll(su));
            }
    };
    {
        // the thread's first FG_KC items: processed one behind the loads; word, scan value and left-neighbour bit stay in
        // registers for the two phases below (k_frame_contours does the same, see there)
        FgMergeItem cur, nxt;
        int pos = im.start;
        if (pos < im.limit) cur = merge_load(wl[pos]);
        int inext = pos + im.step < im.limit ? wl[pos + im.step] : -1;
#pragma unroll
        for (int k = 0; k < FG_KC; k++) {
            kc_x[k] = -1; kc_c[k] = 0; kc_i[k] = 0;
            if (pos < im.limit) {
                const int i2 = (k + 1 < FG_KC && pos + 2 * im.step < im.limit) ? wl[pos + 2 * im.step] : -1;
                if (k + 1 < FG_KC && inext >= 0) nxt = merge_load(inext);
                merge_proc(cur);
                kc_x[k] = cur.idx; kc_c[k] = cur.c; kc_i[k] = cur.id0 | (int)(((cur.cp >> 63) & 1ull) << 31);
                cur = nxt;
                inext = i2;
            }
            pos += im.step;
        }
        if (more) frame_pipeline<FgMergeItem>(wl, im, FG_KC, merge_load, merge_proc);
    }
    __syncthreads();
    FG_PROF(); // 8: merge
    // ---- flatten; last row and strong flag per root
    auto flat_load = [&](int idx) {
            FgWordItem t;
            int y = idx / wq, q = idx - y * wq;
            t.idx = idx;
            t.c = fb[idx];
            t.cp = q > 0 ? fb[idx - 1] : 0ull;
            t.m = mb[idx];
            t.id0 = sf[idx];
            return t;
    };
    auto flat_proc = [&](const FgWordItem &t) {
            int y = t.idx / wq, q = t.idx - y * wq;
            u64 c = t.c & valid_mask(q, w);
            u64 s = c & ~((c << 1) | (t.cp >> 63));
            int id = t.id0;
            for (; s; id++) {
                s &= s - 1;
                int root = lds_find(L, id);
                if (root != id) {
                    L[id] = root;
                    // last row of the component: only runs with nothing below them can hold it (a big
                    // component would otherwise serialise thousands of atomics on one address)
                    if (!((HB[id >> 5] >> (id & 31)) & 1u)) { if (keys_path) atomicMax(&YM[root], y); else atomicMax(&YMg[root], y); }
                }
            }
            // strong pixels, stretch by stretch INSIDE this word (a run that started in an earlier word is
            // id0 - 1; its pieces in later words flag the same root, so nobody walks across words)
            u64 rem = c & t.m ? c : 0ull;
            s = c & ~((c << 1) | (t.cp >> 63));
            while (rem) {
                int b = __ffsll((long long)rem) - 1;
                u64 inv = ~(c >> b);
                int len = inv ? (__ffsll((long long)inv) - 1) : (64 - b);
                u64 seg = (len >= 64 ? ~0ull : ((1ull << len) - 1)) << b;
                rem &= ~seg;
                if (!(t.m & seg)) continue;
                int root = lds_find(L, t.id0 + __popcll(s & upto_bit(b)) - 1);
                atomicOr(&FL[root >> 5], 1u << (root & 31));
            }
    };
    auto cached_item = [&](int k, u64 m) {
        FgWordItem t;
        t.idx = kc_x[k]; t.c = kc_c[k]; t.id0 = kc_i[k] & 0x7fffffff; t.cp = (u64)((unsigned)kc_i[k] >> 31) << 63; t.m = m;
        return t;
    };
    {
        u64 mk[FG_KC]; // the strong bits of the thread's items: the only loads of this phase, all in flight together
#pragma unroll
        for (int k = 0; k < FG_KC; k++) mk[k] = kc_x[k] >= 0 ? mb[kc_x[k]] : 0ull;
#pragma unroll
        for (int k = 0; k < FG_KC; k++) if (kc_x[k] >= 0) flat_proc(cached_item(k, mk[k]));
        if (more) frame_pipeline<FgWordItem>(wl, im, FG_KC, flat_load, flat_proc);
    }
    __syncthreads();
    FG_PROF(); // 9: flatten + strong
    // ---- contour keys of the edge components (their outer borders), keys_path only
    const int NOEDGE = INT_MIN, EDGE_NOSLOT = INT_MIN + 1;
    auto edge_load_early = [&](int idx) { // (items beyond the register copies)
            FgWordItem t;
            int y = idx / wq, q = idx - y * wq;
            t.idx = idx;
            word_pair(fb, idx, q > 0, 0ull, t.cp, t.c);
            t.m = 0;
            t.id0 = sf[idx];
            return t;
    };
    int4 *kg = keys + (size_t)g * key_cap;
    int2 *re = rowext + (size_t)g * slot_cap;
    bool lds_slots = false;
    int n_slots = 0;
    int *SL = X; // (x min, x max) pairs over the dead last-row table
    if (keys_path) {
        // labels and flags for the contour stage / the general kernels (before the table is rewritten below)
        for (int i = threadIdx.x; i < nrun; i += FRAME_THREADS) {
            const int root = L[i];
            Lf[ro + i] = root;
            FLf[ro + i] = (root == i) ? (int)((FL[i >> 5] >> (i & 31)) & 1u) : 0; // root of an edge component
        }
        // a key per root of an edge component.  The thread that holds a root's word knows its row: no per-run row table is written
        // or read (round 4; 14 000 scattered stores per frame in the merge phase, for the ~300 rows this loop wants)
        auto keys_proc = [&](const FgWordItem &t) {
            const int y0 = t.idx / wq, q = t.idx - y0 * wq;
            const u64 c = t.c & valid_mask(q, w);
            const u64 s = c & ~((c << 1) | (t.cp >> 63));
            for (int k = 0, n = __popcll(s); k < n; k++) {
                const int i = t.id0 + k;
                if (L[i] != i || !((FL[i >> 5] >> (i & 31)) & 1u)) continue;
                const int ymax = YM[i], extent = ymax - y0 + 1;
                ROWg[i] = y0; YMg[i] = ymax;               // (what the general contour kernels read, should they take this frame)
                const int base = atomicAdd(&c_slots, extent), ki = atomicAdd(&c_keys, 1);
                if (base + extent > slot_cap || ki >= key_cap) {
                    c_ovf = 1;
                    YM[i] = EDGE_NOSLOT;
                    continue;
                }
                kg[ki] = make_int4(i, extent, y0, base);
                if (extent > BIG_KEY_ROWS) bigkeys[(size_t)g * key_cap + atomicAdd(&c_big, 1)] = ki;
                else if (extent > SMALL_KEY_ROWS) medkeys[(size_t)g * key_cap + atomicAdd(&c_med, 1)] = ki;
                YM[i] = base - y0; // row y of this component lives in slot YM[root] + y
            }
        };
#pragma unroll
        for (int k = 0; k < FG_KC; k++) if (kc_x[k] >= 0) keys_proc(cached_item(k, 0ull));
        if (more) frame_pipeline<FgWordItem>(wl, im, FG_KC, edge_load_early, keys_proc);
        __syncthreads();
        // L[run] := slot offset of its component (or "not an edge run"): one LDS read per stretch below
        for (int i = threadIdx.x; i < nrun; i += FRAME_THREADS) {
            const int r = L[i];
            L[i] = ((FL[r >> 5] >> (r & 31)) & 1u) ? YM[r] : NOEDGE;
        }
        __syncthreads(); // the last-row table is dead: its space takes the row slots
        n_slots = min(c_slots, slot_cap);
        lds_slots = 2 * n_slots <= sizeX && c_ovf == 0;
        if (lds_slots) for (int i = threadIdx.x; i < n_slots; i += FRAME_THREADS) { SL[2 * i] = 0x7fffffff; SL[2 * i + 1] = -1; }
        else for (int i = threadIdx.x; i < n_slots; i += FRAME_THREADS) re[i] = make_int2(0x7fffffff, -1);
        __syncthreads();
        FG_PROF(); // 10: outer keys
    }
    // ---- edge = candidate runs whose component holds a strong pixel (keys_path: and the per-row extremes of the outer borders)
    auto edge_load = [&](int idx) {
            FgWordItem t;
            int y = idx / wq, q = idx - y * wq;
            t.idx = idx;
            t.c = fb[idx];
            t.cp = q > 0 ? fb[idx - 1] : 0ull;
            t.m = 0;
            t.id0 = sf[idx];
            return t;
    };
    auto edge_proc = [&](const FgWordItem &t) {
            int y = t.idx / wq, q = t.idx - y * wq;
            u64 c = t.c & valid_mask(q, w);
            u64 s = c & ~((c << 1) | (t.cp >> 63));
            u64 rem = c, res = 0;
            while (rem) {
                int b = __ffsll((long long)rem) - 1;
                u64 inv = ~(c >> b);
                int len = inv ? (__ffsll((long long)inv) - 1) : (64 - b);
                u64 seg = (len >= 64 ? ~0ull : ((1ull << len) - 1)) << b;
                int root = L[t.id0 + __popcll(s & upto_bit(b)) - 1]; // a stretch continuing from the previous word: id0 - 1
                if ((FL[root >> 5] >> (root & 31)) & 1u) res |= seg;
                rem &= ~seg;
            }
            edge[fo + t.idx] = res;
    };
    auto edge_keys_proc = [&](const FgWordItem &t) {
            int y = t.idx / wq, q = t.idx - y * wq;
            u64 c = t.c & valid_mask(q, w);
            u64 s = c & ~((c << 1) | (t.cp >> 63));
            u64 rem = c, res = 0;
            // a word's stretches mostly widen the same row slot: consecutive updates of one slot are merged in registers
            int aslot = -1, alo = 0, ahi = 0;
            auto flush = [&]() {
                if (aslot < 0) return;
                if (lds_slots) { if (aslot < n_slots) { atomicMin(&SL[2 * aslot], alo); atomicMax(&SL[2 * aslot + 1], ahi); } }
                else slot_update(re, aslot, alo, ahi);
                aslot = -1;
            };
            while (rem) {
                int b = __ffsll((long long)rem) - 1;
                u64 inv = ~(c >> b);
                int len = inv ? (__ffsll((long long)inv) - 1) : (64 - b);
                u64 seg = (len >= 64 ? ~0ull : ((1ull << len) - 1)) << b;
                rem &= ~seg;
                const int v = L[t.id0 + __popcll(s & upto_bit(b)) - 1]; // a stretch continuing from the previous word: id0 - 1
                if (v == NOEDGE) continue;
                res |= seg;
                if (v == EDGE_NOSLOT) continue;
                const int slot = v + y, xs = (q << 6) + b, xe = xs + len - 1;
                if (slot == aslot) { alo = min(alo, xs); ahi = max(ahi, xe); }
                else { flush(); aslot = slot; alo = xs; ahi = xe; }
            }
            flush();
            edge[fo + t.idx] = res;
    };
    if (keys_path) {
#pragma unroll
        for (int k = 0; k < FG_KC; k++) if (kc_x[k] >= 0) edge_keys_proc(cached_item(k, 0ull));
        if (more) frame_pipeline<FgWordItem>(wl, im, FG_KC, edge_load, edge_keys_proc);
        __syncthreads();
        if (lds_slots) for (int i = threadIdx.x; i < n_slots; i += FRAME_THREADS) re[i] = make_int2(SL[2 * i], SL[2 * i + 1]);
        if (threadIdx.x == 0) {
            int *cnt = counters + g * C_COUNT;
            cnt[C_NSLOTS] = c_slots;
            cnt[C_NKEYS] = c_keys < key_cap ? c_keys : key_cap;
            cnt[C_NBIG] = c_big;
            cnt[C_NMED] = c_med;
            if (c_ovf) cnt[C_OVERFLOW] = 1;
        }
        FG_PROF(); // 11: edge bits + outer extremes
        return;
    }
#pragma unroll
    for (int k = 0; k < FG_KC; k++) if (kc_x[k] >= 0) edge_proc(cached_item(k, 0ull));
    if (more) frame_pipeline<FgWordItem>(wl, im, FG_KC, edge_load, edge_proc);
    FG_PROF(); // 10: edge bits
    for (int i = threadIdx.x; i < nrun; i += FRAME_THREADS) {
        int root = L[i];
        Lf[ro + i] = root;
        FLf[ro + i] = (root == i) ? (int)((FL[i >> 5] >> (i & 31)) & 1u) : 0; // root of an edge component
    }
    FG_PROF(); // 11: write-out
}

struct BgMergeItem { int idx, id0, idu; u64 c, cp, u, up, pl; };
struct BgWordItem { int idx, id0; u64 c, cp; };

// position of the k-th (0-based) set bit of s
__device__ __forceinline__ int hole_bit(u64 s, int k) {
    for (; k > 0; k--) s &= s - 1;
    return __ffsll((long long)s) - 1;
}

struct ExtItem { int idx, id0, sbc, sbu, sbd; u64 e, ep, en, c, cp, u, up, d, dp; };

// Contour topology of one frame's edge image.
//  (1) background: 4-connected components of the 0-runs, "touches the frame" flag per root
//      (FLb: 1 = outside, 0 = hole), last row of every hole (YMb), row of every run (ROWb),
//      flattened labels (Lb) -- what k_runs_init(bg), k_runs_merge4_bg, k_runs_flatten(bg) and
//      k_bg_extent compute;
//  (2) contour keys with their row-extent slots -- k_keys;
//  (3) per-row extremes of every key -- k_extremes;
// with the background labels and flags read from LDS in (2) and (3) and the key / slot counters
// kept in LDS.  Same tables, same values (the order of the keys is as arbitrary as before).
__global__ void __launch_bounds__(FRAME_THREADS)
k_frame_contours(RunTabs t, const int *wl_fg, const int *wl_bg, int *counters, int4 *keys, int *bigkeys, int *medkeys, int2 *rowext,
                 int2 *rsa, int h, int w, int key_cap, int slot_cap, int lds_cap, const int *active, int *fallback, int *pass_flags, long long *prof,
                 int lds_n, const int *perm, int dbg, const int *fg_keys) {
    // fg_keys[frame] != 0: k_frame_fg has made the outer-border keys, their row slots and extremes already (counters C_NKEYS /
    // C_NSLOTS / C_NBIG / C_NMED hold its totals): only the hole borders are left to do here
    // developer profile (prof != nullptr): wall-clock ticks (10 ns) at the end of every phase, per frame
    const long long t0 = prof ? wall_clock64() : 0;
    int pk = 0;
    const int g_prof = perm ? perm[blockIdx.x] : (int)blockIdx.x;
#define FRAME_PROF() do { if (prof && threadIdx.x == 0) prof[g_prof * 16 + (pk++)] = wall_clock64() - t0; } while (0)
    const u64 *edge = t.edge;
    const int *scanb = t.scanb;
    int *Lb = t.Lb, *YMb = t.YMb, *FLb = t.FLb, *ROWb = t.ROWb;
    const int run_cap = t.run_cap;
    const int g = perm ? perm[blockIdx.x] : (int)blockIdx.x;
    if (slot_off(active, counters, g)) {
        if (threadIdx.x == 0) fallback[g] = 0;
        return;
    }
    const int wq = LFD_WQ(w);
    const size_t fo = (size_t)g * h * wq, ro = (size_t)g * run_cap;
    const int nwork = counters[g * C_COUNT + C_NBGW], nrun = counters[g * C_COUNT + C_NRUNB];
    const int nwf = counters[g * C_COUNT + C_NFGW], nrunf = counters[g * C_COUNT + C_NRUNF];
    const bool fits = nrun <= lds_cap && nrunf <= lds_cap; // (then k_frame_fg took the frame too: FLf is set)
    const bool fgk = fg_keys != nullptr && fg_keys[g] != 0;
    const int nkeys_fg = fgk ? min(counters[g * C_COUNT + C_NKEYS], key_cap) : 0; // (read before this kernel adds the holes' keys)
    if (threadIdx.x == 0) {
        fallback[g] = fits ? 0 : 1;
        if (!fits) {
            atomicOr(&pass_flags[g], PASS_FLAG_GENERAL);
            if (fgk) { // the general kernels (k_keys) make every key of this frame themselves
                int *cnt = counters + g * C_COUNT;
                cnt[C_NSLOTS] = 0; cnt[C_NKEYS] = 0; cnt[C_NBIG] = 0; cnt[C_NMED] = 0;
            }
        }
    }
    if (!fits) return;
    extern __shared__ int sm_frame[];
    int *L = sm_frame;
    unsigned *FL = (unsigned *)(sm_frame + lds_n);       // root touches the frame (outside)
    unsigned *HB = FL + lds_n / 32;                      // run touches a 0-run of the next row
    unsigned *HL = HB + lds_n / 32;                      // run belongs to a hole
    unsigned *HR = HL + lds_n / 32;                      // run is the root of a hole
    int *YMg = YMb + ro, *ROWg = ROWb + ro;
    int *XSg = t.PAb + ro; // scratch until the hole keys are made: PAb[root] = first column of the hole
    const u64 *fb = edge + fo;
    const int *sb = scanb + fo, *wl = wl_bg + fo;
    __shared__ int c_slots, c_keys, c_big, c_med, c_ovf, c_hovf;
    // holes of the frame, keyed by their root run: (slot of border row 0 minus that row, surrounding component)
    __shared__ int hkey[FRAME_HOLECAP];
    __shared__ int2 hval[FRAME_HOLECAP];
    if (threadIdx.x == 0) {
        const int *cnt = counters + g * C_COUNT;
        c_slots = fgk ? cnt[C_NSLOTS] : 0; c_keys = fgk ? cnt[C_NKEYS] : 0; c_big = fgk ? cnt[C_NBIG] : 0; c_med = fgk ? cnt[C_NMED] : 0;
        c_ovf = 0; c_hovf = 0;
    }
    for (int i = threadIdx.x; i < FRAME_HOLECAP; i += FRAME_THREADS) hkey[i] = -1;
    for (int i = threadIdx.x; i < (nrun + 31) / 32; i += FRAME_THREADS) { FL[i] = 0u; HB[i] = 0u; HL[i] = 0u; HR[i] = 0u; }
    for (int i = threadIdx.x; i < nrun; i += FRAME_THREADS) L[i] = i;
    __syncthreads();
    // ---- row of every run; 4-connectivity between rows y and y-1
    // A compute unit's gathers are what this kernel is short of (~1.5 clocks per lane and load instruction: round 4's
    // attribution runs), so (1) a word and its left neighbour come as ONE 16-byte load, (2) the merge loads six words per
    // item, not ten: "both runs start at column 0" can only hold for the contact at column 0 itself (a contact stretch that
    // begins further right begins behind an edge pixel of one of the two rows, whose run therefore starts there), which needs
    // no table, and (3) the thread keeps its items' word, scan value and left-neighbour bit in registers: the two phases
    // after the merge walk the same items and load nothing (items beyond BG_KC per thread: crowded frames, gathered again).
    constexpr int BG_KC = 10;
    const ItemMap im = item_map(nwork, dbg & 16);
    const bool more = nwork > BG_KC * FRAME_THREADS;
    u64 kc_c[BG_KC];
    int kc_i[BG_KC], kc_x[BG_KC]; // id0 | (pixel to the left is an edge pixel or the frame) << 31, idx
    const int last_bit = (w - 1) & 63;
    unsigned *CL = HR; // during merge + flatten: run reaches the last column of its row (HR is only needed after them)
    auto merge_load = [&](int idx) {
        BgMergeItem t;
        int y = idx / wq, q = idx - y * wq;
        t.idx = idx;
        word_pair(fb, idx, q > 0, ~0ull, t.cp, t.c);
        t.id0 = sb[idx];
        t.u = 0; t.up = 0; t.idu = 0; t.pl = 0;
        if (y > 0) {
            word_pair(fb, idx - wq, q > 0, ~0ull, t.up, t.u);
            t.idu = sb[idx - wq];
            if (q == 0) t.pl = fb[idx - 1]; // last word of the row above
        }
        return t;
    };
    auto merge_proc = [&](const BgMergeItem &t) {
        int y = t.idx / wq, q = t.idx - y * wq;
        u64 vmask = valid_mask(q, w);
        u64 z = ~t.c & vmask;                              // 0-pixels of this word
        u64 s = z & ~((z << 1) | ((~t.cp) >> 63));         // 0-run starts (cp = all ones left of column 0)
        // (no per-run row / last-row tables: only the roots of holes need them, and those are written in the hole phase)
        if (y == 0) return;
        // the run before this row's first one is the last run of the row above: it touches the frame if that row ends in a 0
        if (q == 0 && t.id0 > 0 && !((t.pl >> last_bit) & 1ull)) atomicOr(&CL[(t.id0 - 1) >> 5], 1u << ((t.id0 - 1) & 31));
        u64 zu = ~t.u & vmask;
        u64 su = zu & ~((zu << 1) | ((~t.up) >> 63));
        u64 v = z & zu;
        // first column of every stretch; a stretch continuing from the previous word (both rows 0
        // at the last column of that word) was already joined there
        u64 cont = q > 0 ? ((~t.cp & ~t.up) >> 63) : 0ull;
        u64 st = v & ~((v << 1) | cont);
        // two runs that both start at column 0 touch the frame: each is flagged "outside" on its own, joining them would
        // only build a 1 489-link chain down the left image border
        if (q == 0) st &= ~1ull;
        while (st) {
            int b = __ffsll((long long)st) - 1;
            st &= st - 1;
            int ia = t.id0 + __popcll(s & upto_bit(b)) - 1, ib = t.idu + __popcll(su & upto_bit(b)) - 1;
            atomicOr(&HB[ib >> 5], 1u << (ib & 31));
            if (!(dbg & 8)) lds_union(L, ia, ib);
        }
    };
    {   // the thread's first BG_KC items: processed one behind the loads, kept for the next two phases
        BgMergeItem cur, nxt;
        int pos = im.start;
        if (pos < im.limit) cur = merge_load(wl[pos]);
        int inext = pos + im.step < im.limit ? wl[pos + im.step] : -1;
#pragma unroll
        for (int k = 0; k < BG_KC; k++) {
            kc_x[k] = -1; kc_c[k] = 0; kc_i[k] = 0;
            if (pos < im.limit) {
                const int i2 = (k + 1 < BG_KC && pos + 2 * im.step < im.limit) ? wl[pos + 2 * im.step] : -1;
                if (k + 1 < BG_KC && inext >= 0) nxt = merge_load(inext);
                merge_proc(cur);
                kc_x[k] = cur.idx; kc_c[k] = cur.c; kc_i[k] = cur.id0 | (int)(((cur.cp >> 63) & 1ull) << 31);
                cur = nxt;
                inext = i2;
            }
            pos += im.step;
        }
    }
    if (more) frame_pipeline<BgMergeItem>(wl, im, BG_KC, merge_load, merge_proc);
    __syncthreads();
    FRAME_PROF(); // 0: background merge
    // ---- flatten; outside flag per root
    auto word_load = [&](int idx) { // (items beyond the register copies)
        BgWordItem t;
        int q = idx % wq;
        t.idx = idx;
        word_pair(fb, idx, q > 0, ~0ull, t.cp, t.c);
        t.id0 = sb[idx];
        return t;
    };
    auto cached_item = [&](int k) {
        BgWordItem t;
        t.idx = kc_x[k]; t.c = kc_c[k]; t.id0 = kc_i[k] & 0x7fffffff; t.cp = (u64)((unsigned)kc_i[k] >> 31) << 63;
        return t;
    };
    auto flat_proc = [&](const BgWordItem &t) {
        int y = t.idx / wq, q = t.idx - y * wq;
        u64 z = ~t.c & valid_mask(q, w);
        u64 s = z & ~((z << 1) | ((~t.cp) >> 63));
        int id = t.id0;
        for (; s; id++) {
            int b = __ffsll((long long)s) - 1;
            s &= s - 1;
            int root = lds_find(L, id);
            if (root != id) L[id] = root;
            // touches the frame: first / last row, starts at column 0, or reaches the last column (CL, set during the merge)
            bool flag = (y == 0) || (y == h - 1) || ((q << 6) + b == 0) || ((CL[id >> 5] >> (id & 31)) & 1u);
            if (flag && !((FL[root >> 5] >> (root & 31)) & 1u)) atomicOr(&FL[root >> 5], 1u << (root & 31));
        }
    };
#pragma unroll
    for (int k = 0; k < BG_KC; k++) if (kc_x[k] >= 0) flat_proc(cached_item(k));
    if (more) frame_pipeline<BgWordItem>(wl, im, BG_KC, word_load, flat_proc);
    __syncthreads();
    for (int i = threadIdx.x; i < (nrun + 31) / 32; i += FRAME_THREADS) HR[i] = 0u; // (CL is done: the space is HR's from here on)
    __syncthreads();
    FRAME_PROF(); // 1: flatten
    // ---- last row of every hole; which runs belong to a hole, which are a hole's root
    auto hole_proc = [&](const BgWordItem &t) {
        int y = t.idx / wq, q = t.idx - y * wq;
        u64 z = ~t.c & valid_mask(q, w);
        u64 s = z & ~((z << 1) | ((~t.cp) >> 63));
        int n = __popcll(s);
        for (int k = 0; k < n; k++) {
            int id = t.id0 + k, root = L[id];
            if ((FL[root >> 5] >> (root & 31)) & 1u) continue; // outside
            atomicOr(&HL[id >> 5], 1u << (id & 31));
            if (root == id) {
                atomicOr(&HR[id >> 5], 1u << (id & 31));
                XSg[id] = (q << 6) + hole_bit(s, k); // column of the hole's raster-first pixel
                ROWg[id] = y;                        // row / last row: of a hole's root only (a few hundred per frame, not every run)
                YMg[id] = y;
            }
        }
    };
    auto hole_extent_proc = [&](const BgWordItem &t) { // (after the roots' entries are in place)
        int y = t.idx / wq, q = t.idx - y * wq;
        u64 z = ~t.c & valid_mask(q, w);
        u64 s = z & ~((z << 1) | ((~t.cp) >> 63));
        int n = __popcll(s);
        for (int k = 0; k < n; k++) {
            int id = t.id0 + k;
            if (!((HL[id >> 5] >> (id & 31)) & 1u) || ((HB[id >> 5] >> (id & 31)) & 1u)) continue; // not a hole's run, or one with a run below
            int root = L[id];
            if (root != id) atomicMax(&YMg[root], y);
        }
    };
#pragma unroll
    for (int k = 0; k < BG_KC; k++) if (kc_x[k] >= 0) hole_proc(cached_item(k));
    if (more) frame_pipeline<BgWordItem>(wl, im, BG_KC, word_load, hole_proc);
    __syncthreads(); // (device-scope ordering of the plain stores above and the atomics below is by the memory-side atomic unit: see the note at the read)
#pragma unroll
    for (int k = 0; k < BG_KC; k++) if (kc_x[k] >= 0) hole_extent_proc(cached_item(k));
    if (more) frame_pipeline<BgWordItem>(wl, im, BG_KC, word_load, hole_extent_proc);
    if (prof) { __syncthreads(); FRAME_PROF(); } // 2: hole extents
    for (int i = threadIdx.x; i < nrun; i += FRAME_THREADS) {
        int root = L[i];
        Lb[ro + i] = root;
        FLb[ro + i] = (root == i) ? (int)((FL[i >> 5] >> (i & 31)) & 1u) : 0;
    }
    if (prof) { __syncthreads(); FRAME_PROF(); } // 3: write-out
    // ---- contour keys of the edge components (outer borders): one per root run of an edge component
    const u64 *cb = t.cand + fo;
    const int *sf = t.scanf + fo, *wlf = wl_fg + fo;
    const int *Lfg = t.Lf + ro, *YMfg = t.YMf + ro, *ROWfg = t.ROWf + ro, *FLfg = t.FLf + ro;
    int *SBfg = t.SBf + ro, *SBbg = t.SBb + ro, *PAbg = t.PAb + ro;
    int4 *kg = keys + (size_t)g * key_cap;
    int2 *re = rowext + (size_t)g * slot_cap;
    auto new_key = [&](int id, int ymin, int extent, bool hole) -> int { // slot base, or -1
        int base = atomicAdd(&c_slots, extent);
        int ki = atomicAdd(&c_keys, 1);
        if (base + extent > slot_cap || ki >= key_cap) {
            c_ovf = 1;
            return -1;
        }
        kg[ki] = make_int4(id, extent | (hole ? KEY_HOLE_BIT : 0), ymin, base);
        if (extent > BIG_KEY_ROWS) bigkeys[(size_t)g * key_cap + atomicAdd(&c_big, 1)] = ki;
        else if (extent > SMALL_KEY_ROWS) medkeys[(size_t)g * key_cap + atomicAdd(&c_med, 1)] = ki;
        return base; // the slots themselves are initialised by the whole workgroup, see below
    };
    if (!fgk)
    for (int i0 = threadIdx.x; i0 < nrunf; i0 += 4 * FRAME_THREADS) { // four independent loads in flight per lane
        int lf[4], fl[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int i = i0 + k * FRAME_THREADS;
            lf[k] = i < nrunf ? Lfg[i] : -1;
            fl[k] = i < nrunf ? FLfg[i] : 0;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int i = i0 + k * FRAME_THREADS;
            if (lf[k] != i || !fl[k]) continue;
            int y0 = ROWfg[i];
            SBfg[i] = new_key(i, y0, YMfg[i] - y0 + 1, false);
        }
    }
    __syncthreads(); // hole extents (memory-side atomics), SBf and the hole bits are complete
    const int n_outer_slots = min(c_slots, slot_cap);
    if (!fgk) for (int i = threadIdx.x; i < n_outer_slots; i += FRAME_THREADS) re[i] = make_int2(0x7fffffff, -1);
    FRAME_PROF(); // 4: outer keys
    // ---- rsa[i] = (row-extent slot of candidate run i in its component's key, that component): one load
    // per edge stretch later instead of a chain through Lf / SBf / ROWf
    int2 *RSA = rsa + (size_t)g * FRAME_RUNCAP;
    if (!fgk)
    for (int i0 = threadIdx.x; i0 < nrunf; i0 += 4 * FRAME_THREADS) {
        int A[4], r[4], sbA[4], rA[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int i = i0 + k * FRAME_THREADS;
            A[k] = i < nrunf ? Lfg[i] : 0;
            r[k] = i < nrunf ? ROWfg[i] : 0;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int i = i0 + k * FRAME_THREADS;
            sbA[k] = i < nrunf ? SBfg[A[k]] : -1;
            rA[k] = i < nrunf ? ROWfg[A[k]] : 0;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int i = i0 + k * FRAME_THREADS;
            if (i < nrunf) RSA[i] = make_int2(sbA[k] >= 0 ? sbA[k] + r[k] - rA[k] : -1, A[k]); // (only edge runs are ever looked up)
        }
    }
    if (prof) { __syncthreads(); FRAME_PROF(); } // 5: slot table
    // ---- hole keys: border rows run from the row above a hole's first pixel to the row below its last;
    // its surrounding component is that of the (edge) pixel right above the first pixel.  One lane per
    // 32 background runs of the root bitset: only actual holes cost memory round trips.
    for (int wd = threadIdx.x; wd < (nrun + 31) / 32; wd += FRAME_THREADS) {
        unsigned bits = HR[wd];
        while (bits) {
            int id = (wd << 5) + __ffs((int)bits) - 1;
            bits &= bits - 1;
            int y = ROWg[id], x = XSg[id];
            int ymax = __hip_atomic_load(&YMg[id], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int parent = Lfg[run_id(sf, cb, y - 1, x, 1, wq, w)];
            int base = new_key(id, y - 1, ymax - y + 3, true);
            SBbg[id] = base;
            PAbg[id] = parent; // (overwrites the scratch column)
            // border row yy of this hole lives in slot base + (yy - (y - 1))
            unsigned hs = ((unsigned)id * 2654435761u) >> 22; // 10 bits
            bool put = false;
            for (int probe = 0; probe < 16 && !put; probe++, hs = (hs + 1) & (FRAME_HOLECAP - 1))
                if (atomicCAS(&hkey[hs], -1, id) == -1) {
                    hval[hs] = make_int2(base >= 0 ? base - (y - 1) : INT_MIN, parent);
                    put = true;
                }
            if (!put) c_hovf = 1;
        }
    }
    __syncthreads(); // rsa, the hole table, SBb / PAb of this frame are written
    // fgk: which holes have a component INSIDE them.  The edge pixels next to a hole B belong to the component around it (A) or
    // to components whose own outside is B -- "islands": exactly the components whose raster-first pixel has a run of B to its
    // left (a component next to B that is not A cannot be around B, so B is its exterior).  Only a hole with an island needs
    // the "is this stretch part of A" test in the extremes below -- and with it the stretch's component label, a gather, and
    // the candidate word and scan value that lead to it.  Sky frames: rings without islands almost everywhere.
    unsigned *HI = HB; // (HB is not needed after the hole-extent phase)
    if (fgk) {
        for (int i = threadIdx.x; i < (nrun + 31) / 32; i += FRAME_THREADS) HI[i] = 0u;
        __syncthreads();
        for (int ki = threadIdx.x; ki < nkeys_fg; ki += FRAME_THREADS) {
            const int4 key = kg[ki];                       // an outer-border key made by k_frame_fg: (root run, rows, first row, slot base)
            const int y0 = key.z, x0 = re[key.w].x;        // the first row's left-most pixel: the component's raster-first pixel
            if (x0 <= 0 || x0 >= w) continue;              // column 0: the exterior is the frame
            const int bid = run_id(sb, fb, y0, x0 - 1, 0, wq, w);
            if (bid < 0 || bid >= nrun || !((HL[bid >> 5] >> (bid & 31)) & 1u)) continue;
            const int B = L[bid];
            atomicOr(&HI[B >> 5], 1u << (B & 31));
        }
        __syncthreads();
    }
    // Row extremes are min / max updates of (slot.x, slot.y), two per edge stretch and hole contact: as memory-side atomics they
    // were 40 % of this kernel.  The label table only uses its first nrun entries, so when the frame's slots fit into the rest
    // of it (they do on sky frames: a few thousand slots, 8 bytes each) the updates go to LDS and are copied out once at the end.
    const int n_slots = min(c_slots, slot_cap);
    const int sl0 = fgk ? n_outer_slots : 0;               // first slot this kernel updates (the outer borders' are done)
    int *SL = L + ((nrun + 1) & ~1);                       // (x, y) pairs of slots sl0 .. n_slots - 1; 8-byte aligned
    const bool lds_slots = 2 * (n_slots - sl0) <= lds_n - ((nrun + 1) & ~1) && c_ovf == 0;
    if (lds_slots) {
        for (int i = threadIdx.x; i < n_slots - sl0; i += FRAME_THREADS) { SL[2 * i] = 0x7fffffff; SL[2 * i + 1] = -1; }
    } else
        for (int i = n_outer_slots + threadIdx.x; i < n_slots; i += FRAME_THREADS) re[i] = make_int2(0x7fffffff, -1);
    __syncthreads(); // every slot is initialised before the first update
    FRAME_PROF(); // 6: hole keys
    // ---- per-row extremes: every edge run widens its component's outer-border key, and the
    // hole-border key of every hole it is 4-adjacent to (if its component surrounds that hole)
    const int *ROWbg = ROWb + ro;
    const bool hovf = c_hovf != 0;
    // Slot updates are min / max, so consecutive updates of one slot are merged in registers and sent as
    // one atomic pair: a word's stretches mostly widen the same outer-border row and the same hole
    // (the inside of their own ring); the memory-side atomics are what this phase is short of.
    struct SlotAcc { int slot, lo, hi; };
    auto acc_flush = [&](SlotAcc &a) {
        if (a.slot >= 0 && !(dbg & 4)) {
            if (lds_slots) {
                const int k = a.slot - sl0;
                if ((unsigned)k < (unsigned)(n_slots - sl0)) { atomicMin(&SL[2 * k], a.lo); atomicMax(&SL[2 * k + 1], a.hi); }
            } else slot_update(re, a.slot, a.lo, a.hi);
        }
        a.slot = -1;
    };
    auto acc_add = [&](SlotAcc &a, int slot, int xa, int xb) {
        if (slot == a.slot) {
            a.lo = min(a.lo, xa);
            a.hi = max(a.hi, xb);
            return;
        }
        acc_flush(a);
        a.slot = slot; a.lo = xa; a.hi = xb;
    };
    // A: the edge stretch's component, or -2 = not looked up yet (fgk: fetched by get_A only for a hole with an island inside)
    auto hole_update = [&](SlotAcc &acc, int bid, int &A, auto get_A, int y, int xa, int xb) {
        if (dbg & 2) return;
        if (!((HL[bid >> 5] >> (bid & 31)) & 1u)) return; // a run of the outside
        int B = L[bid];
        const bool check = !fgk || ((HI[B >> 5] >> (B & 31)) & 1u); // no island in B: every edge pixel next to it is the surrounding component's
        if (check && A == -2) A = get_A();
        unsigned hs = ((unsigned)B * 2654435761u) >> 22;
        for (int probe = 0; probe < 16; probe++, hs = (hs + 1) & (FRAME_HOLECAP - 1)) {
            int kk = hkey[hs];
            if (kk == B) {
                int2 v = hval[hs];
                if ((!check || v.y == A) && v.x != INT_MIN) acc_add(acc, v.x + y, xa, xb);
                return;
            }
            if (kk == -1) break;
        }
        if (hovf && (!check || PAbg[B] == A) && SBbg[B] >= 0) acc_add(acc, SBbg[B] + (y - (ROWbg[B] - 1)), xa, xb); // table was full
    };
    frame_pipeline<ExtItem>(
        wlf, nwf,
        [&](int idx) {
            ExtItem k;
            int y = idx / wq, q = idx - y * wq;
            k.idx = idx;
            word_pair(fb, idx, q > 0, ~0ull, k.ep, k.e); // "edge" left of column 0: no background run continues from there
            k.en = q + 1 < wq ? fb[idx + 1] : 0ull;
            k.c = 0; k.cp = 0; k.id0 = 0;
            if (!fgk) { // (fgk: the stretch's candidate run is only needed next to a hole with an island: fetched there)
                word_pair(cb, idx, q > 0, 0ull, k.cp, k.c);
                k.id0 = sf[idx];
            }
            k.sbc = sb[idx];
            k.u = ~0ull; k.up = ~0ull; k.sbu = 0; k.d = ~0ull; k.dp = ~0ull; k.sbd = 0;
            if (y > 0) {
                word_pair(fb, idx - wq, q > 0, ~0ull, k.up, k.u);
                k.sbu = sb[idx - wq];
            }
            if (y + 1 < h) {
                word_pair(fb, idx + wq, q > 0, ~0ull, k.dp, k.d);
                k.sbd = sb[idx + wq];
            }
            return k;
        },
        [&](const ExtItem &k) {
            int y = k.idx / wq, q = k.idx - y * wq;
            u64 vmask = valid_mask(q, w);
            u64 e = k.e & vmask, c = k.c & vmask;
            u64 epe = q > 0 ? k.ep : 0ull; // edge bits of the previous word (none left of column 0)
            u64 se = e & ~((e << 1) | (epe >> 63)), sc = c & ~((c << 1) | (k.cp >> 63));
            u64 z = ~e & vmask;
            u64 s0 = z & ~((z << 1) | ((~k.ep) >> 63)); // 0-run starts of this word
            // Every maximal stretch of edge pixels INSIDE this word is handled here, whether or not its
            // run started in an earlier word or goes on in the next one: the slot updates are min / max,
            // so the pieces of a run add up to the run, and no lane ever walks across words.
            SlotAcc outer = {-1, 0, 0}, hole = {-1, 0, 0};
            u64 rem = e;
            while (rem) {
                int b = __ffsll((long long)rem) - 1;
                u64 inv = ~(e >> b);
                int len = inv ? (__ffsll((long long)inv) - 1) : (64 - b);
                u64 seg = (len >= 64 ? ~0ull : ((1ull << len) - 1)) << b;
                rem &= ~seg;
                int xs = (q << 6) + b, xe = xs + len - 1;
                int A = -2;
                auto get_A = [&]() -> int { // fgk: the component of this stretch's candidate run (label table of k_frame_fg, in memory)
                    u64 cc, ccp;
                    word_pair(cb, k.idx, q > 0, 0ull, ccp, cc);
                    cc &= vmask;
                    const u64 scc = cc & ~((cc << 1) | (ccp >> 63));
                    const int f = sf[k.idx] + __popcll(scc & upto_bit(b)) - 1; // a stretch continuing from the previous word: id0 - 1
                    return (f >= 0 && f < nrunf) ? Lfg[f] : -1;
                };
                if (!fgk) {
                    int fid = k.id0 + __popcll(sc & upto_bit(b)) - 1; // a stretch continuing from the previous word: id0 - 1
                    if (fid < 0 || fid >= nrunf) continue;
                    int2 ra = (dbg & 1) ? make_int2(fid & 1023, 0) : RSA[fid];
                    A = ra.y;
                    if (ra.x >= 0) acc_add(outer, ra.x, xs, xe);
                }
                // same-row neighbours: the 0-pixel before the run and the one after it
                bool starts = (se >> b) & 1ull;
                bool ends = (b + len < 64) || q + 1 >= wq || !(k.en & 1ull);
                if (starts && xs > 0) hole_update(hole, b ? k.sbc + __popcll(s0 & upto_bit(b - 1)) - 1 : k.sbc - 1, A, get_A, y, xs, xs);
                if (ends && xe < w - 1)
                    hole_update(hole, (b + len < 64) ? k.sbc + __popcll(s0 & upto_bit(b + len)) - 1 : k.sbc + __popcll(s0), A, get_A, y, xe, xe);
                // rows above and below: 0-runs overlapping [xs, xe]
                for (int dy = -1; dy <= 1; dy += 2) {
                    int yy = y + dy;
                    if (yy < 0 || yy >= h) continue;
                    u64 ow = dy < 0 ? k.u : k.d, owp = dy < 0 ? k.up : k.dp;
                    int sbo = dy < 0 ? k.sbu : k.sbd;
                    u64 zz = ~ow & vmask;
                    u64 s0o = zz & ~((zz << 1) | ((~owp) >> 63));
                    u64 ov = zz & seg;
                    while (ov) {
                        int bx = __ffsll((long long)ov) - 1;
                        u64 inv2 = ~(ov >> bx);
                        int l2 = inv2 ? (__ffsll((long long)inv2) - 1) : (64 - bx);
                        hole_update(hole, sbo + __popcll(s0o & upto_bit(bx)) - 1, A, get_A, y, (q << 6) + bx, (q << 6) + bx + l2 - 1);
                        ov &= ~((l2 >= 64 ? ~0ull : ((1ull << l2) - 1)) << bx);
                    }
                }
            }
            acc_flush(outer);
            acc_flush(hole);
        });
    __syncthreads();
    if (lds_slots)
        for (int i = threadIdx.x; i < n_slots - sl0; i += FRAME_THREADS) re[sl0 + i] = make_int2(SL[2 * i], SL[2 * i + 1]);
    FRAME_PROF(); // 7: extremes
    if (threadIdx.x == 0) {
        int *cnt = counters + g * C_COUNT;
        cnt[C_NSLOTS] = c_slots;
        cnt[C_NKEYS] = c_keys < key_cap ? c_keys : key_cap;
        cnt[C_NBIG] = c_big;
        cnt[C_NMED] = c_med;
        if (c_ovf) cnt[C_OVERFLOW] = 1;
    }
}
