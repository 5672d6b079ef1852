// k_bz2.h -- bzip2 decompression on the device: the step in front of the hot path on real archives (SDSS serves frames as
// frame-*.fits.bz2; the reference decompresses every one with `bunzip2` before it reads it, detecttrails.py:81-109, at ~0.4 s
// per frame and core).  A frame is ~14 independent blocks of 900 kB; a chunk of 256 frames is ~3 600 blocks.
//
//   k_bz2_magics     every 48-bit block / end-of-stream magic of every file, at any bit offset      (bandwidth, trivial)
//   k_bz2_huff       a WAVE per block: header, coding tables, Huffman + RUNA/RUNB + move-to-front   (one wave, 900 000 symbols one
//                    -> the block's BWT column L                                                      after the other: bound by the
//                                                                                                     CU's scalar ALU at 14 waves)
//   k_bz2_sort       a workgroup per block: stable counting sort of L -> tt[q] = (T[q] << 8) | byte  (LDS histograms, ballots)
//   k_bz2_walk       inverse BWT as list ranking: ~3 500 splitters per block walk to the next        (random 4-byte loads: HBM's
//                    splitter leaving their bytes in scratch, one thread orders the splitters,        random-access rate)
//                    the scratch stretches are copied to their places
//   k_bz2_rle_tiles  the run-length layer (4 equal bytes + count) in 32 kB tiles: every ~32-byte     (LDS scan)
//   k_bz2_rle_blocks stretch parsed under both possible entry states, the functions composed by a
//                    scan inside the tile, tile by tile inside the block
//   k_bz2_offsets    output offset of every block inside its file
//   k_bz2_expand     writes the file's bytes through an LDS window; every stretch's share of its block's CRC
//   k_bz2_crc_check  (moved to the block's end with x^(8 len) mod P, xor-ed) against the stored one
// All integer / byte work: bit-exact by construction, and every block's stored CRC is checked on the device.
#pragma once
#include "common.h"
#include "bz2_core.h"

#define BZ_LSTRIDE 900608     // bytes per block of the L / pre-RLE buffers (>= 900 000 + a staging buffer of 256, a multiple of 256)
#define BZ_TSTRIDE 900096     // words per block of tt
#define BZ_SEL_STRIDE 21248    // per block: 18 002 selectors, then the six tables' perm[] (3 096 bytes)
#define BZ_SPLIT_LOG 8        // a splitter every 256 positions of tt
#define BZ_MAX_SPLIT 3520     // 900 000 / 256 + head, rounded up
#define BZ_MARK_CAP 256       // magics kept per file (a level-1 file of 12.6 MB has 127 blocks)

struct BzBlockDesc {
    uint64_t start_bit, end_bit; // of the block's magic / of the next magic, relative to the file's first word
    uint64_t word_off, nwords;   // the file inside the device's copy of the compressed bytes
    int file, max_block;
};

// ---- magics -----------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_bz2_magics(const uint32_t *comp, const uint64_t *word_off, const uint64_t *nbytes, int *nfound, u64 *marks) {
    const int f = blockIdx.y;
    const uint64_t nb = nbytes[f];
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; // this thread: magics that START in bytes 8 i .. 8 i + 7
    if (i * 8 >= nb) return;
    const uint32_t *w = comp + word_off[f] + i * 2; // (the copy is padded with zero words: reading 16 bytes is safe)
    const u64 hi = ((u64)bz_bswap32(w[0]) << 32) | bz_bswap32(w[1]), lo = ((u64)bz_bswap32(w[2]) << 32) | bz_bswap32(w[3]);
    for (int o = 0; o < 64; o++) {
        const u64 v = (o ? ((hi << o) | (lo >> (64 - o))) : hi) >> 16;
        const bool blk = v == 0x314159265359ull, eos = v == 0x177245385090ull;
        if (blk || eos) {
            const u64 bit = i * 64 + (u64)o;
            if (bit + 48 > nb * 8) continue;
            const int k = atomicAdd(&nfound[f], 1);
            if (k < BZ_MARK_CAP) marks[(size_t)f * BZ_MARK_CAP + k] = (bit << 1) | (eos ? 1ull : 0ull);
        }
    }
}

// ---- Huffman + MTF: a wave per block ----------------------------------------------------------------------------------------
struct BzDevIO {
    uint8_t *len;
    int *limit, *base, *min_len;
    uint16_t *perm, *fast;
    uint8_t *sel;
    uint8_t *L;
    int flushed, cnt, lane;
    uint32_t vz;    // zero, in a vector register the compiler knows nothing about: `x + vz` keeps wave-uniform arithmetic on the
                    // vector ALUs (four per CU) instead of the one scalar ALU all waves of a CU share, which is what bounds this kernel
    uint32_t stage; // output bytes in flight, NEWEST FIRST: byte p of the 256 (lane p / 4, bits 8 (p % 4) ..) is the (p + 1)-th youngest
    uint32_t list;  // the move-to-front list (byte values), lane l holds entries 4 l .. 4 l + 3, entry j in bits 8 j ..
    __device__ __forceinline__ void mtf_begin() { list = 0; }
    __device__ __forceinline__ void mtf_add(int k, uint32_t b) {
        list |= lane == (k >> 2) ? b << (8 * (k & 3)) : 0u;
    }
    __device__ __forceinline__ uint32_t mtf_head() { return __builtin_amdgcn_readfirstlane(list) & 0xffu; }
    // One ordinary symbol: the byte at position nn >= 1 of the list moves to the front and is appended to the output.  Both are
    // "shift everything below by one byte and put v in front": a byte shift inside each lane with the top byte of lane l - 1
    // coming in at the bottom (DPP wave_shr; lane 0 has no lane below it and takes v from the instruction's `old` operand).
    __device__ __forceinline__ void symbol(int nn, int q) { // q = nn >> 2 (the caller has it from the table entry); cnt is the caller's to count

        const uint32_t wq = (uint32_t)__builtin_amdgcn_readlane((int)list, q);
        const uint32_t r8 = (((uint32_t)nn + vz) & 3u) << 3;          // (vector registers from here on)
        const uint32_t v24 = (wq >> r8) << 24;                         // the byte, in the top byte
        const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp((int)v24, (int)list, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
        const uint32_t sh = (list << 8) | (prev >> 24);
        const uint32_t mr = ~(0xffffff00u << r8);                      // bytes 0 .. r of lane q
        const uint32_t m = lane < q ? 0xffffffffu : (lane == q ? mr : 0u);
        list = (sh & m) | (list & ~m);
        const uint32_t sprev = (uint32_t)__builtin_amdgcn_update_dpp((int)v24, (int)stage, 0x138, 0xf, 0xf, false);
        stage = (stage << 8) | (sprev >> 24);
    }
    __device__ __forceinline__ void emit(uint32_t b) { // (a byte that is not a list move: runs)
        const uint32_t v24 = (b + vz) << 24;
        const uint32_t sprev = (uint32_t)__builtin_amdgcn_update_dpp((int)v24, (int)stage, 0x138, 0xf, 0xf, false);
        stage = (stage << 8) | (sprev >> 24);
        cnt++;
    }
    // The cnt <= 256 bytes in flight, oldest first, to L[flushed ..].  Every lane stores all four of its bytes whatever cnt is: the
    // ones that hold nothing go to L[flushed + p], p >= cnt -- beyond the block's current end, overwritten by what comes next (the
    // buffer has 256 bytes of slack) -- so that there is no lane-dependent branch here: one inside the symbol loop makes the
    // compiler treat the whole loop as divergent and keep its counters in vector registers.
    __device__ __forceinline__ void flush() {
        for (int j = 0; j < 4; j++) {
            const int p = 4 * lane + j;
            L[flushed + (p < cnt ? cnt - 1 - p : p)] = (uint8_t)(stage >> (8 * j));
        }
        flushed += cnt;
        cnt = 0;
    }
    __device__ __forceinline__ void emit_run(uint32_t b, int n) {
        if (n >= 64) { // long: straight to memory (whole groups of 64 bytes: up to 63 beyond the run's end, see flush)
            flush();
            for (int k0 = 0; k0 < n; k0 += 64) L[flushed + k0 + lane] = (uint8_t)b;
            flushed += n;
            return;
        }
        if (cnt + n > 256) flush();
        for (int k = 0; k < n; k++) emit(b);
    }
    __device__ __forceinline__ void flush_tail() { flush(); }
    __device__ __forceinline__ int emitted() const { return flushed + cnt; }
    __device__ __forceinline__ void build_fast(int t, int mn) {
        __syncthreads(); // (one wave: orders the LDS traffic of the table building before the fast table overwrites `len`)
        for (int k = 0; k < BZ_FAST_SIZE / 64; k++) {
            const uint32_t x = (uint32_t)(lane + 64 * k);
            // (the kernel's own encoding: position in the move-to-front list = symbol - 1, mod 512, in front of the code length:
            // RUNA 511, RUNB 0, end of block n_in_use: the exceptions are `entry's position - 1 >= n_in_use - 1`, unsigned)
            const uint32_t e0 = bz_fast_entry(*this, t, mn, x);
            fast[t * BZ_FAST_SIZE + x] = (uint16_t)(e0 ? ((((e0 >> 4) - 1u) & 0x1ffu) << 4) | (e0 & 15u) : 0u);
        }
        __syncthreads();
    }
};

// The coded symbols of a block, decoded by one wave.  Every lane looks up the code that would start at ITS bit offset of a 64-bit
// window (one LDS read for 64 candidates); the wave then follows the chain of code lengths through the window with v_readlane:
// ~8 symbols per window on incompressible data, more on sky.  The move-to-front list and the output staging live in registers
// (BzDevIO).  A table switch (every 50 symbols) or a code longer than BZ_FAST_BITS ends a window early.
__device__ __forceinline__ int bz_dev_symbols(BzDevIO &io, const uint32_t *w, uint64_t nwords, uint64_t bit0, const BzHeader &hd,
                                              int max_block, uint64_t &bit_end) {
    const int lane = io.lane;
    // (what the header code hands over is wave-uniform, but the compiler has lost track of that across its lane-parallel table set-up)
    const int eob = __builtin_amdgcn_readfirstlane(hd.n_in_use) + 1, n_sel = __builtin_amdgcn_readfirstlane(hd.n_sel);
    const uint32_t b0lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)bit0), b0hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(bit0 >> 32));
    const uint64_t bit0u = ((uint64_t)b0hi << 32) | b0lo;
    uint64_t base = bit0u >> 5; // first word held in wv
    int bp = (int)(bit0u & 31); // the cursor, in bits from word `base`
    io.cnt = __builtin_amdgcn_readfirstlane(io.cnt);
    io.flushed = __builtin_amdgcn_readfirstlane(io.flushed);
    auto loadw = [&](uint64_t b) {
        const uint64_t i = b + (uint64_t)lane;
        return i < nwords ? bz_bswap32(w[i]) : 0u;
    };
    uint32_t wv = loadw(base);
    int group_no = -1, group_pos = 0, t = 0;
    int run_n = 0, run_len = 0;
    const int lsh = lane & 31;
    const bool upper = lane >= 32;
    int st = -1, bad = 0;
    uint32_t thr = (uint32_t)(eob - 2); // (unsigned)(sym - 2) >= thr: sym is RUNA / RUNB / end-of-block (or anything, while a run is open)
    while (st < 0) {
        if (group_pos == 0) {
            group_no++;
            if (group_no >= n_sel) { st = BZ_E_DATA; break; }
            group_pos = BZ_GROUP_SYMS;
            t = __builtin_amdgcn_readfirstlane((int)io.sel[group_no]);
        }
        int k0 = bp >> 5;
        if (k0 > 59) { // keep four whole words ahead of the cursor inside wv
            base += (uint64_t)k0;
            bp &= 31;
            wv = loadw(base);
            k0 = 0;
        }
        const int c = bp & 31;
        const uint32_t a0 = __builtin_amdgcn_readlane(wv, k0), a1 = __builtin_amdgcn_readlane(wv, k0 + 1);
        const uint32_t a2 = __builtin_amdgcn_readlane(wv, k0 + 2), a3 = __builtin_amdgcn_readlane(wv, k0 + 3);
        const u64 A = ((u64)a0 << 32) | a1, B = ((u64)a2 << 32) | a3;
        const u64 hi = c ? ((A << c) | (B >> (64 - c))) : A; // the 64 bits at the cursor
        const uint32_t nx = (uint32_t)((B << c) >> 32);       // and the 32 after them
        const uint32_t h0 = (uint32_t)(hi >> 32), h1 = (uint32_t)hi;
        // lane l: the BZ_FAST_BITS bits starting l bits after the cursor
        const u64 pair = upper ? (((u64)h1 << 32) | nx) : (((u64)h0 << 32) | h1);
        const uint32_t prefix = (uint32_t)((pair << lsh) >> (64 - BZ_FAST_BITS));
        const uint32_t ev = io.fast[t * BZ_FAST_SIZE + prefix];
        if (io.cnt > 256 - 64) io.flush(); // (a window holds at most 64 codes: no check per symbol)
        int pos = 0;
        bool done = false;
        for (;;) { // stretches of ordinary symbols, an exception between two of them (errors are collected in `bad`, looked at once per window)
            uint32_t e;
            // position in the window and symbols left under this table in ONE register: bits 0 .. 7 = pos (a code is at most 9 bits
            // here: < 73), bits 8 .. 14 = 64 - group_pos; every ordinary symbol adds its length + 256, and the loop is over when pos
            // reaches 64 (bit 6) or the count reaches 64 (bit 14)
            uint32_t key = (uint32_t)pos | ((uint32_t)(64 - group_pos) << 8);
            for (;;) { // the ordinary symbols: nothing in here touches what the exceptions change (run state, thr, flushed, cnt)
                e = (uint32_t)__builtin_amdgcn_readlane((int)ev, (int)(key & 63u));
                const int nn = (int)(e >> 4);
                if (__builtin_expect((uint32_t)(nn - 1) >= thr, 0)) break;
                key += (e & 15u) + 256u;
                io.symbol(nn, (int)(e >> 6)); // (a block that grows beyond its level's size is caught when a buffer is flushed)
                if (key & 0x40c0u) break; // the table's 50 symbols are used up, or the window is
            }
            pos = (int)(key & 0xffu);
            {
                const int left = 64 - (int)((key >> 8) & 0x7fu);
                io.cnt += group_pos - left; // one byte per ordinary symbol
                group_pos = left;
            }
            if (key & 0x40c0u) break; // (which of the two exits it was: this one cannot hold when an exception ended the loop)
            // the exceptions, all behind that one test: a code longer than the look-up covers (e = 0), RUNA / RUNB, the end-of-block
            // symbol, and any symbol while a run is being collected (thr = 0 then)
            int sym = (int)(((e >> 4) + 1u) & 0x1ffu);
            int len = (int)(e & 15u);
            if (e == 0u) {
                BzBits br;
                const uint64_t at = base * 32 + (uint64_t)(bp + pos);
                br.init(w, nwords, at);
                sym = bz_slow_symbol(io, br, t);
                len = (int)(br.pos - at);
                if (sym < 0) { bad = 1; sym = 2; }
            }
            pos += len;
            if (sym <= 1) {
                if (run_n == 0) { run_n = 1; run_len = 0; thr = 0; }
                if (run_n >= 2 * 1024 * 1024) { bad = 1; run_n = 1; }
                run_len += run_n << sym;
                run_n <<= 1;
            } else {
                if (run_n != 0) {
                    if (run_len > max_block - io.emitted()) { bad = 1; run_len = 0; }
                    io.emit_run(io.mtf_head(), run_len);
                    run_n = 0;
                    thr = (uint32_t)(eob - 2);
                    if (io.cnt > 256 - 64) io.flush();
                }
                if (sym == eob) { done = true; break; }
                io.symbol(sym - 1, (sym - 1) >> 2);
                io.cnt++;
            }
            if (--group_pos == 0 || pos >= 64) break;
        }
        if (done) {
            bit_end = base * 32 + (uint64_t)(bp + pos);
            st = BZ_OK;
        }
        if (bad) st = BZ_E_DATA;
        bp += pos;
        if (io.flushed > max_block) st = BZ_E_DATA;
    }
    return st;
}

__global__ void __launch_bounds__(64)
k_bz2_huff(const uint32_t *comp, const BzBlockDesc *desc, BzBlockInfo *info, uint8_t *Lbuf, uint8_t *selbuf, int nblocks) {
    const int b = blockIdx.x;
    if (b >= nblocks) return;
    __shared__ __attribute__((aligned(16))) uint16_t s_fast[BZ_MAX_GROUPS * BZ_FAST_SIZE];
    __shared__ int s_limit[BZ_MAX_GROUPS * BZ_NLEN], s_base[BZ_MAX_GROUPS * BZ_NLEN], s_min[8];
    const BzBlockDesc d = desc[b];
    BzDevIO io;
    io.len = (uint8_t *)s_fast; // code lengths: needed only until the tables exist
    io.fast = s_fast;
    io.perm = (uint16_t *)(selbuf + (size_t)b * BZ_SEL_STRIDE + 18048); // (only the bit-by-bit path and the table set-up read it)
    io.limit = s_limit;
    io.base = s_base;
    io.min_len = s_min;
    io.sel = selbuf + (size_t)b * BZ_SEL_STRIDE;
    io.L = Lbuf + (size_t)b * BZ_LSTRIDE;
    io.flushed = 0;
    io.cnt = 0;
    io.stage = 0;
    asm volatile("v_mov_b32 %0, 0" : "=v"(io.vz));
    io.list = 0;
    io.lane = threadIdx.x;
    BzBlockInfo bi;
    BzBits br;
    const uint32_t *w = comp + d.word_off;
    br.init(w, d.nwords, d.start_bit);
    BzHeader hd;
    if (bz_read_header(io, br, bi, hd) == BZ_OK) {
        __syncthreads(); // (selectors written to memory by this wave are read back below)
        uint64_t bit_end = 0;
        int st = bz_dev_symbols(io, w, d.nwords, br.pos, hd, d.max_block, bit_end);
        io.flush_tail();
        bi.nblock = io.emitted();
        if (st == BZ_OK && bi.nblock > d.max_block) st = BZ_E_DATA;
        if (st == BZ_OK && (bi.orig_ptr < 0 || bi.orig_ptr >= bi.nblock)) st = BZ_E_ORIGPTR;
        if (st == BZ_OK && d.end_bit && bit_end != d.end_bit) st = BZ_E_LENGTH;
        bi.status = st;
    }
    if (threadIdx.x == 0) info[b] = bi;
}

// ---- tt: stable counting sort of the BWT column ------------------------------------------------------------------------------
// tt[q] = (i << 8) | L[i] for the i-th byte of L, placed at q = (bytes smaller than L[i]) + (equal bytes before i): following
// q -> tt[q] >> 8 from orig_ptr yields the block's bytes in order, each step's byte in the low 8 bits of the entry just read.
__global__ void __launch_bounds__(1024)
k_bz2_sort(const BzBlockInfo *info, const uint8_t *Lbuf, uint32_t *ttbuf) {
    const int b = blockIdx.x;
    if (info[b].status != BZ_OK) return;
    const int n = info[b].nblock;
    const uint8_t *L = Lbuf + (size_t)b * BZ_LSTRIDE;
    uint32_t *tt = ttbuf + (size_t)b * BZ_TSTRIDE;
    __shared__ int hist[16][256];
    __shared__ int tot[256];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int seg = (((n + 15) / 16) + 63) & ~63, s0 = min(n, wv * seg), s1 = min(n, s0 + seg);
    for (int k = tid; k < 16 * 256; k += 1024) (&hist[0][0])[k] = 0;
    __syncthreads();
    for (int i = s0 + lane; i < s1; i += 64) atomicAdd(&hist[wv][L[i]], 1);
    __syncthreads();
    if (tid < 256) {
        int s = 0;
        for (int w = 0; w < 16; w++) s += hist[w][tid];
        tot[tid] = s;
    }
    __syncthreads();
    if (wv == 0) { // exclusive scan of the 256 totals: four per lane, then across the wave
        int a0 = tot[4 * lane], a1 = tot[4 * lane + 1], a2 = tot[4 * lane + 2], a3 = tot[4 * lane + 3];
        int inc = a0 + a1 + a2 + a3;
        const int loc = inc;
        for (int off = 1; off < 64; off <<= 1) {
            int t = __shfl_up(inc, off);
            if (lane >= off) inc += t;
        }
        int run = inc - loc;
        tot[4 * lane] = run; run += a0;
        tot[4 * lane + 1] = run; run += a1;
        tot[4 * lane + 2] = run; run += a2;
        tot[4 * lane + 3] = run;
    }
    __syncthreads();
    if (tid < 256) {
        int run = tot[tid];
        for (int w = 0; w < 16; w++) {
            const int t = hist[w][tid];
            hist[w][tid] = run;
            run += t;
        }
    }
    __syncthreads();
    for (int i0 = s0; i0 < s1; i0 += 64) {
        const int i = i0 + lane;
        const bool valid = i < s1;
        const uint32_t v = valid ? L[i] : 0u;
        u64 m = __ballot(valid);
#pragma unroll
        for (int bit = 0; bit < 8; bit++) {
            const bool one = (v >> bit) & 1u;
            const u64 bal = __ballot(one);
            m &= one ? bal : ~bal;
        }
        const int rank = __popcll(m & ((1ull << lane) - 1ull)), peers = __popcll(m);
        int base = 0;
        if (valid) {
            base = hist[wv][v];
            tt[base + rank] = ((uint32_t)i << 8) | v;
        }
        if (valid && rank == peers - 1) hist[wv][v] = base + peers;
    }
}

// ---- inverse BWT by list ranking --------------------------------------------------------------------------------------------
struct BzPacker { // consecutive bytes to memory, as word stores wherever a whole aligned word belongs to this writer
    uint8_t *cur;
    uint32_t acc;
    int have, count;
    __device__ __forceinline__ void init(uint8_t *at) { cur = at; acc = 0; have = 0; count = 0; }
    __device__ __forceinline__ void put(uint32_t b) {
        const int sh = (int)((uintptr_t)cur & 3);
        acc |= b << (8 * sh);
        have++;
        if (sh == 3) {
            if (have == 4) *(uint32_t *)(cur - 3) = acc;
            else for (int j = 4 - have; j < 4; j++) cur[j - 3] = (uint8_t)(acc >> (8 * j));
            acc = 0;
            have = 0;
        }
        cur++;
        count++;
    }
    __device__ __forceinline__ void finish() { // (the bytes still in acc end right before cur)
        for (int j = 0; j < have; j++) {
            uint8_t *a = cur - have + j;
            *a = (uint8_t)(acc >> (8 * (int)((uintptr_t)a & 3)));
        }
        have = 0;
        acc = 0;
    }
};

#define BZ_CHAINS 4      // splitters per thread, walked side by side (independent loads in flight)
#define BZ_SEG_CAP 1024  // bytes a stretch writes to its scratch area (stretches average 256 steps; e^-4 = 2 % of them are longer;
                         // 2048 costs 3.6 MB more per block and is no faster)
// The permutation is walked ONCE: every splitter's stretch (to the next splitter) leaves its bytes in a scratch area of its own and
// its length; one thread then puts the stretches in chain order; the scratch areas are copied to their places with coalesced
// reads.  (A stretch longer than its scratch area remembers where it was after BZ_SEG_CAP steps and walks the rest again.)
// A permutation whose cycle through orig_ptr is shorter than the block (periodic data) yields that cycle's bytes over and over.
__global__ void __launch_bounds__(1024)
k_bz2_walk(BzBlockInfo *info, const uint32_t *ttbuf, uint8_t *prebuf, uint8_t *segbuf) {
    const int b = blockIdx.x;
    if (info[b].status != BZ_OK) return;
    const int n = info[b].nblock;
    const uint32_t orig = (uint32_t)info[b].orig_ptr;
    const uint32_t *tt = ttbuf + (size_t)b * BZ_TSTRIDE;
    uint8_t *pre = prebuf + (size_t)b * BZ_LSTRIDE;
    uint8_t *seg = segbuf + (size_t)b * BZ_MAX_SPLIT * BZ_SEG_CAP;
    __shared__ int nxt[BZ_MAX_SPLIT], slen[BZ_MAX_SPLIT], spos[BZ_MAX_SPLIT];
    __shared__ uint32_t qcap[BZ_MAX_SPLIT];
    __shared__ int s_period;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int G = 1 << BZ_SPLIT_LOG;
    const int Ks = (n + G - 1) >> BZ_SPLIT_LOG; // regular splitters 0 .. Ks - 1 at q = s G; splitter Ks is orig_ptr, the head of the chain
    for (int k = tid; k <= Ks; k += 1024) spos[k] = -1;
    uint32_t q[BZ_CHAINS];
    int len[BZ_CHAINS];
    bool on[BZ_CHAINS];
    BzPacker pk[BZ_CHAINS];
#pragma unroll
    for (int k = 0; k < BZ_CHAINS; k++) {
        const int s = tid + 1024 * k;
        on[k] = s <= Ks;
        q[k] = s == Ks ? orig : ((uint32_t)s << BZ_SPLIT_LOG);
        len[k] = 0;
        pk[k].init(seg + (size_t)min(s, BZ_MAX_SPLIT - 1) * BZ_SEG_CAP);
    }
    for (int step = 0; step < n; step++) { // (a stretch ends at the next splitter: ~256 steps on average)
        bool any = false;
#pragma unroll
        for (int k = 0; k < BZ_CHAINS; k++)
            if (on[k]) {
                const uint32_t e = tt[q[k]];
                if (len[k] < BZ_SEG_CAP) pk[k].put(e & 0xffu);
                q[k] = e >> 8;
                len[k]++;
                if (len[k] == BZ_SEG_CAP) qcap[tid + 1024 * k] = q[k];
                if ((q[k] & (uint32_t)(G - 1)) == 0u || q[k] == orig) {
                    const int s = tid + 1024 * k;
                    nxt[s] = q[k] == orig ? Ks : (int)(q[k] >> BZ_SPLIT_LOG);
                    slen[s] = len[k];
                    on[k] = false;
                } else any = true;
            }
        if (!any) break;
    }
#pragma unroll
    for (int k = 0; k < BZ_CHAINS; k++) pk[k].finish();
    __syncthreads();
    if (tid == 0) { // the splitters in chain order: where in the output each one's stretch begins
        int pos = 0, j = Ks;
        while (pos < n) {
            if (spos[j] != -1) break; // back at the head before n steps: the cycle through orig_ptr has `pos` elements
            spos[j] = pos;
            pos += slen[j];
            j = nxt[j];
        }
        s_period = pos; // n, or the length of the cycle through orig_ptr (the stretches of a cycle add up to its length: never more than n)
        if (pos > n) info[b].status = BZ_E_CYCLE;
    }
    __syncthreads();
    const int period = s_period;
    if (period > n) return;
    // stretches to their places: a wave per stretch, 64 bytes per step
    for (int s = wv; s <= Ks; s += 16) {
        const int p0 = spos[s];
        if (p0 < 0) continue;
        const int cnt = min(slen[s], BZ_SEG_CAP);
        const uint8_t *src = seg + (size_t)s * BZ_SEG_CAP;
        for (int k = lane; k < cnt; k += 64) pre[p0 + k] = src[k];
    }
    // the rest of the long ones
    for (int s = tid; s <= Ks; s += 1024) {
        if (spos[s] < 0 || slen[s] <= BZ_SEG_CAP) continue;
        uint32_t qq = qcap[s];
        for (int c = BZ_SEG_CAP; c < slen[s]; c++) {
            const uint32_t e = tt[qq];
            pre[spos[s] + c] = (uint8_t)e;
            qq = e >> 8;
        }
    }
    if (period < n) { // periodic: the first `period` bytes again and again
        __syncthreads();
        for (int i = period + tid; i < n; i += 1024) pre[i] = pre[i % period];
    }
}

// ---- run-length layer ----------------------------------------------------------------------------------------------------------
// bzip2's first stage: after four equal bytes the next byte is a COUNT of further repeats.  A block's bytes are cut into tiles of
// 32 kB (a workgroup each) and stretches of ~32 bytes (a thread each).  A stretch starts at a position whose byte differs from
// the one before it, so the only thing it inherits is whether its first byte is the count of a run of four that ended exactly
// there (c = 1) or not (c = 0): every stretch is parsed under both assumptions, the resulting functions c -> (c', bytes written)
// are composed by a scan inside the tile, tile by tile inside the block (k_bz2_rle_blocks), and the expansion then knows every
// stretch's true entry state and output offset.  Tiles are staged in LDS with coalesced loads (a thread's stretch is strided
// 36 bytes there: 9 words, no bank conflicts) and the output leaves through an LDS window with aligned word stores.
#define BZ_RT 32                      // nominal bytes per thread
#define BZ_TILE (1024 * BZ_RT)        // bytes per tile
#define BZ_MAX_TILES 28               // 900 000 / 32 768, rounded up
#define BZ_OUTW (36 * 1024)           // output window of a tile in LDS (what does not fit is stored directly)

struct BzRle { int c_out, size; };
struct BzTile { // a tile of the pre-run-length bytes in LDS
    const uint32_t *sm;
    const uint8_t *pre;
    int base, n;
    __device__ __forceinline__ uint32_t get(int p) const { // 0 <= p < n
        const int rel = p - base;
        if (rel >= 0 && rel < BZ_TILE) return (sm[(rel >> 5) * 9 + ((rel & 31) >> 2)] >> ((rel & 3) << 3)) & 0xffu;
        return pre[p];
    }
};
__device__ __forceinline__ void bz_tile_stage(uint32_t *sm, const uint8_t *pre, int base, int n, int tid) {
    for (int j = tid; j < BZ_TILE / 16; j += 1024) {
        const int p = base + 16 * j;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (p < n) v = *(const uint4 *)(pre + p); // (the buffers are padded beyond n: whole 16-byte pieces can be read)
        uint32_t *d = sm + (j >> 1) * 9 + (j & 1) * 4;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
}
__device__ __forceinline__ int bz_tile_boundary(const BzTile &t, int from) { // first position >= from where a run of equal bytes starts
    int p = min(t.n, from);
    while (p > 0 && p < t.n && t.get(p) == t.get(p - 1)) p++;
    return p;
}
__device__ __forceinline__ BzRle bz_rle_parse(const BzTile &t, int s, int e, int c) {
    BzRle r;
    r.size = 0;
    int p = s;
    if (c) {
        if (p >= e) { r.c_out = 1; return r; } // nothing here: the count is further on
        r.size += (int)t.get(p++);
    }
    int k = 0;
    uint32_t v = 0x100u;
    while (p < e) {
        const uint32_t x = t.get(p++);
        if (x == v) {
            k++;
            r.size++;
            if (k == 4) {
                if (p >= e) { r.c_out = 1; return r; }
                r.size += (int)t.get(p++);
                k = 0;
                v = 0x100u;
            }
        } else { v = x; k = 1; r.size++; }
    }
    r.c_out = 0;
    return r;
}

// per tile: every stretch's transfer function, scanned; meta = what precedes each stretch inside its tile (for c = 0 / 1 at the
// tile's start: c', bytes), tile_fn = the whole tile
__global__ void __launch_bounds__(1024)
k_bz2_rle_tiles(const BzBlockInfo *info, const uint8_t *prebuf, int4 *meta, int4 *tile_fn) {
    const int b = blockIdx.y, tile = blockIdx.x;
    if (info[b].status != BZ_OK) return;
    const int n = info[b].nblock, base = tile * BZ_TILE;
    if (base >= n) return;
    __shared__ uint32_t sm[1024 * 9];
    __shared__ int bnd[1025];
    __shared__ int f_c0[2][1024], f_c1[2][1024], f_s0[2][1024], f_s1[2][1024];
    const int t = threadIdx.x;
    BzTile T;
    T.sm = sm; T.pre = prebuf + (size_t)b * BZ_LSTRIDE; T.base = base; T.n = n;
    bz_tile_stage(sm, T.pre, base, n, t);
    __syncthreads();
    bnd[t] = bz_tile_boundary(T, base + BZ_RT * t);
    if (t == 1023) bnd[1024] = bz_tile_boundary(T, base + BZ_TILE);
    __syncthreads();
    const int s = bnd[t], e = bnd[t + 1];
    const BzRle r0 = bz_rle_parse(T, s, e, 0), r1 = bz_rle_parse(T, s, e, 1);
    int c0 = r0.c_out, c1 = r1.c_out, z0 = r0.size, z1 = r1.size, cur = 0;
    f_c0[0][t] = c0; f_c1[0][t] = c1; f_s0[0][t] = z0; f_s1[0][t] = z1;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        if (t >= off) { // (left part, then this one): the left part's outcome selects which of this part's two rows applies
            const int lc0 = f_c0[cur][t - off], lc1 = f_c1[cur][t - off], ls0 = f_s0[cur][t - off], ls1 = f_s1[cur][t - off];
            const int n_c0 = lc0 ? c1 : c0, n_s0 = ls0 + (lc0 ? z1 : z0);
            const int n_c1 = lc1 ? c1 : c0, n_s1 = ls1 + (lc1 ? z1 : z0);
            c0 = n_c0; c1 = n_c1; z0 = n_s0; z1 = n_s1;
        }
        cur ^= 1;
        f_c0[cur][t] = c0; f_c1[cur][t] = c1; f_s0[cur][t] = z0; f_s1[cur][t] = z1;
        __syncthreads();
    }
    const size_t ti = (size_t)b * BZ_MAX_TILES + tile;
    meta[ti * 1024 + t] = t ? make_int4(f_c0[cur][t - 1], f_c1[cur][t - 1], f_s0[cur][t - 1], f_s1[cur][t - 1]) : make_int4(0, 1, 0, 0);
    if (t == 1023) tile_fn[ti] = make_int4(c0, c1, z0, z1);
}

// per block: the tiles in order -> every tile's entry state and output offset inside the block, and the block's size
__global__ void k_bz2_rle_blocks(const BzBlockInfo *info, const int4 *tile_fn, int2 *tile_in, int *blk_size, int nblocks) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblocks) return;
    int c = 0, off = 0;
    if (info[b].status == BZ_OK) {
        const int nt = (info[b].nblock + BZ_TILE - 1) / BZ_TILE;
        for (int k = 0; k < nt; k++) {
            const size_t ti = (size_t)b * BZ_MAX_TILES + k;
            tile_in[ti] = make_int2(c, off);
            const int4 f = tile_fn[ti];
            off += c ? f.w : f.z;
            c = c ? f.y : f.x;
        }
    }
    blk_size[b] = off;
}

__global__ void k_bz2_offsets(const BzBlockInfo *info, const int *blk_size, const int *file_first, int nfiles, u64 out_cap,
                              u64 *blk_off, u64 *out_len, int *file_status) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= nfiles) return;
    u64 off = 0;
    int st = file_status[f];
    for (int b = file_first[f]; b < file_first[f + 1]; b++) {
        if (info[b].status != BZ_OK && st == BZ_OK) st = info[b].status;
        blk_off[b] = off;
        off += (u64)blk_size[b];
    }
    if (st == BZ_OK && off > out_cap) st = BZ_E_SIZE;
    out_len[f] = off;
    file_status[f] = st;
}

struct BzCrcPow { uint32_t pw[24]; }; // x^(8 * 2^k) mod P

__device__ __forceinline__ uint32_t bz_gf_mul(uint32_t a, uint32_t b) { // a * b mod P, bit i = coefficient of x^i
    uint32_t r = 0;
    for (int i = 31; i >= 0; i--) {
        r = (r << 1) ^ ((r & 0x80000000u) ? 0x04c11db7u : 0u);
        if ((b >> i) & 1u) r ^= a;
    }
    return r;
}
__device__ __forceinline__ uint32_t bz_x_pow8(const BzCrcPow &pows, int nbytes) { // x^(8 nbytes) mod P
    uint32_t pw = 1u;
    for (int k = 0; k < 24; k++)
        if ((nbytes >> k) & 1) pw = bz_gf_mul(pw, pows.pw[k]);
    return pw;
}

// the file's bytes + every stretch's share of its block's CRC (register started at 0, moved to the block's end: times
// x^(8 * bytes after it) mod P; the shares are xor-ed together in blk_crc, k_bz2_crc_check compares)
__global__ void __launch_bounds__(1024)
k_bz2_expand(const BzBlockInfo *info, const BzBlockDesc *desc, const uint8_t *prebuf, const int4 *meta, const int4 *tile_fn,
             const int2 *tile_in, const int *blk_size, const u64 *blk_off, uint8_t *out, u64 out_cap, const int *file_status,
             uint32_t *blk_crc, BzCrcPow pows) {
    const int b = blockIdx.y, tile = blockIdx.x;
    const int f = desc[b].file;
    if (file_status[f] != BZ_OK) return;
    const int n = info[b].nblock, base = tile * BZ_TILE;
    if (base >= n) return;
    __shared__ uint32_t sm[1024 * 9];
    __shared__ uint32_t ow[BZ_OUTW / 4 + 1];
    __shared__ int bnd[1025];
    __shared__ uint32_t tab[256];
    const int t = threadIdx.x;
    BzTile T;
    T.sm = sm; T.pre = prebuf + (size_t)b * BZ_LSTRIDE; T.base = base; T.n = n;
    bz_tile_stage(sm, T.pre, base, n, t);
    if (t < 256) tab[t] = bz_crc_table_entry((uint32_t)t);
    __syncthreads();
    bnd[t] = bz_tile_boundary(T, base + BZ_RT * t);
    if (t == 1023) bnd[1024] = bz_tile_boundary(T, base + BZ_TILE);
    __syncthreads();
    const size_t ti = (size_t)b * BZ_MAX_TILES + tile;
    const int2 tin = tile_in[ti];
    const int4 tf = tile_fn[ti], pm = meta[ti * 1024 + t];
    const int tile_bytes = tin.x ? tf.w : tf.z;                 // what this tile writes
    const int c_in = tin.x ? pm.y : pm.x;
    const int off = tin.y + (tin.x ? pm.w : pm.z);              // this stretch's first output byte inside the block
    const int total = blk_size[b];
    uint8_t *dst = out + (size_t)f * out_cap + blk_off[b];      // the block's output
    uint8_t *owb = (uint8_t *)ow;
    uint32_t crc = 0;
    int o = off - tin.y;                                        // position inside the tile's output
    auto put = [&](uint32_t v) {
        if (o < BZ_OUTW) owb[o] = (uint8_t)v;
        else dst[tin.y + o] = (uint8_t)v;
        o++;
        crc = (crc << 8) ^ tab[(crc >> 24) ^ v];
    };
    int p = bnd[t];
    const int e = bnd[t + 1];
    if (c_in && p < e) { // the count of the run of four that ended right before this stretch
        const uint32_t v = T.get(p - 1);
        const int rep = (int)T.get(p++);
        for (int k = 0; k < rep; k++) put(v);
    }
    {
        int k = 0;
        uint32_t v = 0x100u;
        while (p < e) {
            const uint32_t x = T.get(p++);
            if (x == v) {
                k++;
                put(x);
                if (k == 4) {
                    if (p >= e) break; // (its count is the next stretch's first byte)
                    const int rep = (int)T.get(p++);
                    for (int j = 0; j < rep; j++) put(x);
                    k = 0;
                    v = 0x100u;
                }
            } else { v = x; k = 1; put(x); }
        }
    }
    const int after = total - (tin.y + o);
    uint32_t part = bz_gf_mul(crc, bz_x_pow8(pows, after));
    if (t == 0 && tile == 0) part ^= bz_gf_mul(0xffffffffu, bz_x_pow8(pows, total)); // the initial register value, moved across the whole block
    for (int sh = 32; sh > 0; sh >>= 1) part ^= __shfl_xor(part, sh);
    if ((t & 63) == 0 && part) atomicXor(&blk_crc[b], part);
    __syncthreads();
    // the window -> memory: bytes up to the first aligned address, whole words, the last bytes
    const int cnt = min(tile_bytes, BZ_OUTW);
    uint8_t *d0 = dst + tin.y;
    const int head = min(cnt, (int)((4 - ((uintptr_t)d0 & 3)) & 3));
    if (t < head) d0[t] = owb[t];
    const int nwords = (cnt - head) >> 2;
    const int sh8 = (head & 3) << 3;
    uint32_t *dw = (uint32_t *)(d0 + head);
    for (int k = t; k < nwords; k += 1024) {
        const int w = (head >> 2) + k; // (head < 4: w = k)
        const uint32_t lo = ow[w], hi = ow[w + 1];
        dw[k] = sh8 ? (lo >> sh8) | (hi << (32 - sh8)) : lo;
    }
    const int done = head + 4 * nwords;
    if (t < cnt - done) d0[done + t] = owb[done + t];
}

__global__ void k_bz2_crc_check(const BzBlockInfo *info, const BzBlockDesc *desc, const uint32_t *blk_crc, int *file_status, int nblocks) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblocks) return;
    const int f = desc[b].file;
    if (file_status[f] != BZ_OK) return;
    if (~blk_crc[b] != info[b].crc) atomicCAS(&file_status[f], BZ_OK, BZ_E_CRC);
}

// first `head` bytes of every decoded file, side by side (for the caller's header parsing)
__global__ void k_bz2_heads(const uint8_t *out, u64 out_cap, const u64 *out_len, uint8_t *heads, u64 head) {
    const int f = blockIdx.y;
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= head) return;
    heads[(size_t)f * head + i] = i < out_len[f] ? out[(size_t)f * out_cap + i] : 0;
}
