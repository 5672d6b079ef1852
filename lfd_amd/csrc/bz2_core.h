// bz2_core.h -- the sequential part of a bzip2 block decoder: block header, coding tables, Huffman + run-length (RUNA / RUNB) +
// move-to-front decoding of one block into the last column of its Burrows-Wheeler matrix.  The reference decompresses a frame's
// .fits.bz2 twin before it reads it (detecttrails.py:81-109: `bunzip2` into $FITS_DUMP); the file format is that of bzip2 1.0.x
// (block layout, coding-table construction and the order of the validity checks follow its published decoder).
//
// One function body serves two builds: the device kernel k_bz2_huff (k_bz2.h: a wave per block, every value wave-uniform, the
// move-to-front list spread over the lanes) and a plain C++ build for tests (tools/bz2_core_check.cpp).  `IO` supplies the table
// storage and the three operations that differ: mtf_front, emit, emit_run.
#pragma once
#include <stdint.h>

#ifdef __HIPCC__
#define BZ_HD __host__ __device__ __forceinline__
#else
#define BZ_HD inline
#endif

#define BZ_MAX_SELECTORS 18002
#define BZ_MAX_ALPHA 258
#define BZ_MAX_CODE_LEN 20
#define BZ_NLEN 24            // entries of limit[] / base[] per table
#define BZ_FAST_BITS 9
#define BZ_FAST_SIZE (1 << BZ_FAST_BITS)
#define BZ_GROUP_SYMS 50
#define BZ_MAX_GROUPS 6

// status of a block / a file: 0 = decoded and its CRC is right; anything else: the caller decodes the file on the host
enum {
    BZ_OK = 0,
    BZ_E_MAGIC = 1,      // no block magic where one was expected
    BZ_E_RANDOMISED = 2, // a randomised block (bzip2 < 0.9.5): valid, not handled here
    BZ_E_HEADER = 3,     // symbol map / groups / selectors / code lengths out of range
    BZ_E_DATA = 4,       // a code that no table entry matches, run too long, block larger than its level allows
    BZ_E_LENGTH = 5,     // the block does not end where the next magic starts
    BZ_E_ORIGPTR = 6,
    BZ_E_CRC = 7,
    BZ_E_CYCLE = 8,      // the BWT permutation does not come back to its start after exactly nblock steps
    BZ_E_SIZE = 9,       // output larger than the caller's buffer
    BZ_E_STREAM = 10     // not a sequence of "BZh1-9" streams (no end mark, bytes after it, a magic where none belongs, ...)
};

struct BzBlockInfo {
    int status;
    int nblock;          // bytes of the BWT column
    int orig_ptr;
    uint32_t crc;        // the block's stored CRC
};

static BZ_HD uint32_t bz_bswap32(uint32_t x) { return (x >> 24) | ((x >> 8) & 0xff00u) | ((x << 8) & 0xff0000u) | (x << 24); }

// Big-endian bit reader over 32-bit words (the file's bytes as they lie in memory).  `buf` holds `have` unread bits, left-aligned.
struct BzBits {
    const uint32_t *w;
    uint64_t nwords, next, buf, pos;
    int have;
    BZ_HD void init(const uint32_t *words, uint64_t n_words, uint64_t bit) {
        w = words; nwords = n_words; pos = bit;
        next = bit >> 5;
        const int sh = (int)(bit & 31);
        const uint32_t x = next < nwords ? bz_bswap32(w[next]) : 0u;
        next++;
        buf = (uint64_t)x << (32 + sh);
        have = 32 - sh;
    }
    BZ_HD void refill() {
        const uint32_t x = next < nwords ? bz_bswap32(w[next]) : 0u;
        next++;
        buf |= (uint64_t)x << (32 - have);
        have += 32;
    }
    BZ_HD uint32_t peek(int n) { // 1 <= n <= 32
        if (have < n) refill();
        return (uint32_t)(buf >> (64 - n));
    }
    BZ_HD void skip(int n) { buf <<= n; have -= n; pos += (uint64_t)n; }
    BZ_HD uint32_t get(int n) { const uint32_t v = peek(n); skip(n); return v; }
};

// limit / base / perm of one coding table from its code lengths, as bzip2 builds them (canonical codes: by length, then by symbol)
template <class IO> BZ_HD void bz_make_tables(IO &io, int t, int alpha, int &min_len, int &max_len) {
    int mn = 32, mx = 0;
    for (int i = 0; i < alpha; i++) {
        const int l = io.len[t * BZ_MAX_ALPHA + i];
        if (l > mx) mx = l;
        if (l < mn) mn = l;
    }
    int *limit = io.limit + t * BZ_NLEN, *base = io.base + t * BZ_NLEN;
    int pp = 0;
    for (int i = mn; i <= mx; i++)
        for (int j = 0; j < alpha; j++)
            if (io.len[t * BZ_MAX_ALPHA + j] == i) io.perm[t * BZ_MAX_ALPHA + pp++] = (uint16_t)j;
    for (int i = 0; i < BZ_NLEN; i++) base[i] = 0;
    for (int i = 0; i < alpha; i++) base[io.len[t * BZ_MAX_ALPHA + i] + 1]++;
    for (int i = 1; i < BZ_NLEN; i++) base[i] += base[i - 1];
    for (int i = 0; i < BZ_NLEN; i++) limit[i] = 0;
    int vec = 0;
    for (int i = mn; i <= mx; i++) {
        vec += base[i + 1] - base[i];
        limit[i] = vec - 1;
        vec <<= 1;
    }
    for (int i = mn + 1; i <= mx; i++) base[i] = ((limit[i - 1] + 1) << 1) - base[i];
    min_len = mn;
    max_len = mx;
}

// What the decoder does with the BZ_FAST_BITS bits `x` in front of it under table t: (symbol << 4 | code length) if a code of at
// most BZ_FAST_BITS bits matches, 0 otherwise (longer code, or no code at all: the bit-by-bit path decides).
template <class IO> BZ_HD uint16_t bz_fast_entry(const IO &io, int t, int min_len, uint32_t x) {
    const int *limit = io.limit + t * BZ_NLEN, *base = io.base + t * BZ_NLEN;
    for (int zn = min_len; zn <= BZ_FAST_BITS; zn++) {
        const int zvec = (int)(x >> (BZ_FAST_BITS - zn));
        if (zvec <= limit[zn]) {
            const int k = zvec - base[zn];
            if (k < 0 || k >= BZ_MAX_ALPHA) return 0;
            return (uint16_t)((io.perm[t * BZ_MAX_ALPHA + k] << 4) | zn);
        }
    }
    return 0;
}

// Decodes the block whose magic starts at bit `start` of the word stream.  Returns BZ_OK or an error; fills `info`.
//   IO members used: uint8_t *len (6 x 258, scratch while the tables are built; may alias fast), int *limit, *base (6 x 24),
//   uint16_t *perm (6 x 258), uint16_t *fast (6 x 512), uint8_t *sel (18002), int *min_len (6);
//   void mtf_begin(), mtf_add(int k, uint32_t byte) (entry k of the initial list); uint32_t mtf_front(int nn) (byte value at
//   position nn, moved to the front); uint32_t mtf_head(); void emit(uint32_t byte); void emit_run(uint32_t byte, int n);
//   void build_fast(int t, int min_len) (fills fast[t] with bz_fast_entry); int emitted().
struct BzHeader { int n_in_use, n_groups, n_sel; };

// the block's header up to and including its coding tables; `br` is left at the first coded symbol
template <class IO> BZ_HD int bz_read_header(IO &io, BzBits &br, BzBlockInfo &info, BzHeader &hd) {
    info.status = BZ_E_MAGIC; info.nblock = 0; info.orig_ptr = 0; info.crc = 0;
    if (br.get(24) != 0x314159u || br.get(24) != 0x265359u) return BZ_E_MAGIC;
    info.crc = br.get(32);
    if (br.get(1)) return info.status = BZ_E_RANDOMISED;
    const int orig = (int)br.get(24);
    info.orig_ptr = orig;
    // symbol map
    const uint32_t used16 = br.get(16);
    int n_in_use = 0;
    io.mtf_begin();
    for (int i = 0; i < 16; i++) {
        if (!((used16 >> (15 - i)) & 1u)) continue;
        const uint32_t m = br.get(16);
        for (int j = 0; j < 16; j++)
            if ((m >> (15 - j)) & 1u) {
                io.mtf_add(n_in_use, (uint32_t)(i * 16 + j)); // the list starts as the used byte values in ascending order
                n_in_use++;
            }
    }
    if (n_in_use == 0) return info.status = BZ_E_HEADER;
    const int alpha = n_in_use + 2;
    const int n_groups = (int)br.get(3);
    if (n_groups < 2 || n_groups > BZ_MAX_GROUPS) return info.status = BZ_E_HEADER;
    const int n_sel = (int)br.get(15);
    if (n_sel < 1 || n_sel > BZ_MAX_SELECTORS) return info.status = BZ_E_HEADER;
    {
        uint32_t pos = 0x543210u; // the selectors' own move-to-front list, a nibble per entry
        for (int i = 0; i < n_sel; i++) {
            int j = 0;
            while (br.get(1)) {
                j++;
                if (j >= n_groups) return info.status = BZ_E_HEADER;
            }
            const uint32_t tmp = (pos >> (4 * j)) & 15u;
            const uint32_t low = pos & ((1u << (4 * j)) - 1u);
            pos = (pos & ~((1u << (4 * j + 4)) - 1u)) | (low << 4) | tmp;
            io.sel[i] = (uint8_t)tmp;
        }
    }
    for (int t = 0; t < n_groups; t++) {
        int curr = (int)br.get(5);
        for (int i = 0; i < alpha; i++) {
            for (;;) {
                if (curr < 1 || curr > BZ_MAX_CODE_LEN) return info.status = BZ_E_HEADER;
                if (!br.get(1)) break;
                if (br.get(1)) curr--; else curr++;
            }
            io.len[t * BZ_MAX_ALPHA + i] = (uint8_t)curr;
        }
    }
    for (int t = 0; t < n_groups; t++) {
        int mx;
        bz_make_tables(io, t, alpha, io.min_len[t], mx);
    }
    for (int t = 0; t < n_groups; t++) io.build_fast(t, io.min_len[t]); // (after every table is made: fast may alias len)
    hd.n_in_use = n_in_use; hd.n_groups = n_groups; hd.n_sel = n_sel;
    return info.status = BZ_OK;
}

// one symbol the long way (codes longer than BZ_FAST_BITS, or no code at all): bit by bit against limit[]; -1 = no such code
template <class IO> BZ_HD int bz_slow_symbol(const IO &io, BzBits &br, int t) {
    int zn = io.min_len[t];
    if (zn < BZ_FAST_BITS + 1) zn = BZ_FAST_BITS + 1;
    const int *limit = io.limit + t * BZ_NLEN, *base = io.base + t * BZ_NLEN;
    int zvec = (int)br.get(zn);
    for (;;) {
        if (zn > BZ_MAX_CODE_LEN) return -1;
        if (zvec <= limit[zn]) break;
        zn++;
        zvec = (zvec << 1) | (int)br.get(1);
    }
    const int k = zvec - base[zn];
    if (k < 0 || k >= BZ_MAX_ALPHA) return -1;
    return io.perm[t * BZ_MAX_ALPHA + k];
}

// the coded symbols, one after the other (the reference loop: the device kernel has its own, k_bz2.h)
template <class IO> BZ_HD int bz_decode_symbols(IO &io, BzBits &br, const BzHeader &hd, int max_block, BzBlockInfo &info) {
    const int n_in_use = hd.n_in_use, n_sel = hd.n_sel;
    const int eob = n_in_use + 1;
    int group_no = -1, group_pos = 0, t = 0;
    int run_n = 0, run_len = 0; // a run of RUNA / RUNB symbols being collected: its length so far (minus one) and the next weight
    for (;;) {
        if (group_pos == 0) {
            group_no++;
            if (group_no >= n_sel) return info.status = BZ_E_DATA;
            group_pos = BZ_GROUP_SYMS;
            t = io.sel[group_no];
        }
        group_pos--;
        int sym;
        const uint16_t e = io.fast[t * BZ_FAST_SIZE + br.peek(BZ_FAST_BITS)];
        if (e) {
            br.skip(e & 15);
            sym = e >> 4;
        } else {
            sym = bz_slow_symbol(io, br, t);
            if (sym < 0) return info.status = BZ_E_DATA;
        }
        if (sym <= 1) { // RUNA / RUNB: bijective base-2 digits of a run of the byte at the front of the list
            if (run_n == 0) { run_n = 1; run_len = 0; }
            if (run_n >= 2 * 1024 * 1024) return info.status = BZ_E_DATA;
            run_len += run_n << sym;
            run_n <<= 1;
            continue;
        }
        if (run_n) {
            if (run_len > max_block - io.emitted()) return info.status = BZ_E_DATA;
            io.emit_run(io.mtf_head(), run_len);
            run_n = 0;
        }
        if (sym == eob) break;
        if (io.emitted() >= max_block) return info.status = BZ_E_DATA;
        io.emit(io.mtf_front(sym - 1));
    }
    return info.status = BZ_OK;
}

template <class IO> BZ_HD int bz_decode_block(IO &io, const uint32_t *words, uint64_t nwords, uint64_t start, uint64_t end_bit,
                                              int max_block, BzBlockInfo &info) {
    BzBits br;
    br.init(words, nwords, start);
    BzHeader hd;
    if (bz_read_header(io, br, info, hd)) return info.status;
    if (bz_decode_symbols(io, br, hd, max_block, info)) return info.status;
    info.nblock = io.emitted();
    if (info.orig_ptr < 0 || info.orig_ptr >= info.nblock) return info.status = BZ_E_ORIGPTR;
    if (end_bit && br.pos != end_bit) return info.status = BZ_E_LENGTH;
    return info.status = BZ_OK;
}

// CRC of bzip2: polynomial 0x04c11db7, most significant bit first, initial value and final xor 0xffffffff
static BZ_HD uint32_t bz_crc_table_entry(uint32_t i) {
    uint32_t c = i << 24;
    for (int k = 0; k < 8; k++) c = (c & 0x80000000u) ? (c << 1) ^ 0x04c11db7u : (c << 1);
    return c;
}
