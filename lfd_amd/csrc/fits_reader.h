// fits_reader.h -- host-side FITS ingest of the drop-in: the two file layouts the detection path opens, read by native
// threads straight into the buffers the GPU call takes (no interpreter lock, no per-frame Python objects).
//
// The reference reads a frame with fitsio.read (detecttrails.py:113: cfitsio reads the data unit, swaps the bytes, numpy
// copies) and the photoObj table with fitsio column reads + a Python loop of math.ceil per object (removestars.py:96-130).
// Here the data unit of a BITPIX = -32 image goes into its slot of a page-locked buffer AS IT IS IN THE FILE (big-endian;
// the device swaps), and the six photoObj columns remove_stars uses land in the padded lfdmi_catalog arrays.  Files this
// fast path does not cover (other BITPIX, BSCALE / BZERO, scaled table columns, compressed files) are reported, not
// guessed at: the caller's general reader takes them.  Included by lfdmi.hip (host code only).
#pragma once
#include <errno.h>
#include <fcntl.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "../../include/lfdmi.h"

namespace lfd_fits {

constexpr size_t BLOCK = 2880;

// Offset just past the header that starts at `start` (2880-byte blocks of 80-byte cards up to the END card), or 0 if the END
// card is not inside buf[0 .. len).
static size_t header_end(const char *buf, size_t len, size_t start) {
    for (size_t o = start; o + 80 <= len; o += 80)
        if (memcmp(buf + o, "END     ", 8) == 0) {
            bool blank = true;
            for (int k = 8; k < 80 && blank; k++) blank = buf[o + k] == ' ';
            if (blank) return start + ((o - start) / BLOCK + 1) * BLOCK;
        }
    return 0;
}

// value field (columns 11-80) of card `key` in header hdr[0 .. len): false when the card is absent
static bool card(const char *hdr, size_t len, const char *key, std::string *val) {
    char k8[9];
    snprintf(k8, sizeof k8, "%-8s", key);
    for (size_t o = 0; o + 80 <= len; o += 80)
        if (memcmp(hdr + o, k8, 8) == 0 && hdr[o + 8] == '=' && hdr[o + 9] == ' ') {
            val->assign(hdr + o + 10, 70);
            return true;
        }
    return false;
}
static bool card_int(const char *hdr, size_t len, const char *key, long long *out) {
    std::string v;
    if (!card(hdr, len, key, &v)) return false;
    size_t slash = v.find('/');
    if (slash != std::string::npos) v.resize(slash);
    char *end = nullptr;
    long long x = strtoll(v.c_str(), &end, 10);
    if (end == v.c_str()) return false;
    while (*end == ' ') end++;
    if (*end != 0) return false; // (a float or a string: not an integer card)
    *out = x;
    return true;
}
static bool card_double(const char *hdr, size_t len, const char *key, double *out) {
    std::string v;
    if (!card(hdr, len, key, &v)) return false;
    size_t slash = v.find('/');
    if (slash != std::string::npos) v.resize(slash);
    for (auto &c : v) if (c == 'D' || c == 'd') c = 'E';
    char *end = nullptr;
    double x = strtod(v.c_str(), &end);
    if (end == v.c_str()) return false;
    *out = x;
    return true;
}
// quoted string value, trailing blanks dropped ('' inside the quotes is not expected in the cards read here)
static bool card_str(const char *hdr, size_t len, const char *key, std::string *out) {
    std::string v;
    if (!card(hdr, len, key, &v)) return false;
    size_t a = v.find('\'');
    if (a == std::string::npos) return false;
    size_t b = v.find('\'', a + 1);
    if (b == std::string::npos) return false;
    *out = v.substr(a + 1, b - a - 1);
    while (!out->empty() && out->back() == ' ') out->pop_back();
    return true;
}

static bool read_fully(int fd, void *dst, size_t bytes, off_t off) {
    char *p = (char *)dst;
    while (bytes) {
        ssize_t r = pread(fd, p, bytes, off);
        if (r < 0 && errno == EINTR) continue;
        if (r <= 0) return false;
        p += r; off += r; bytes -= (size_t)r;
    }
    return true;
}

template <class F> static void parallel_for(int n, int threads, F f) {
    if (threads < 1) threads = 1;
    if (threads > n) threads = n;
    std::atomic<int> next{0};
    auto work = [&] { for (int i; (i = next.fetch_add(1)) < n;) f(i); };
    std::vector<std::thread> th;
    for (int t = 1; t < threads; t++) th.emplace_back(work);
    work();
    for (auto &x : th) x.join();
}

static inline float be_f32(const unsigned char *p) {
    uint32_t u = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3];
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static inline long long be_int(const unsigned char *p, int size) {
    unsigned long long u = 0;
    for (int k = 0; k < size; k++) u = (u << 8) | p[k];
    if (size == 1) return (long long)u; // TFORM B: unsigned byte
    const int sh = 64 - 8 * size;
    return (long long)(u << sh) >> sh;   // sign extension of I / J / K
}

} // namespace lfd_fits

extern "C" int lfdmi_fits_read_frames(const char *const *paths, int n, int h, int w, void *dst, int threads, int32_t *status,
                                      char *hdr, int hdr_cap, int32_t *hdr_len) {
    using namespace lfd_fits;
    if (!paths || !dst || !status || n < 0 || h <= 0 || w <= 0 || (hdr && (hdr_cap < 80 || !hdr_len))) return LFDMI_ERR_ARG;
    const size_t frame_bytes = (size_t)h * w * 4;
    parallel_for(n, threads, [&](int i) {
        status[i] = -1;
        if (hdr_len) hdr_len[i] = 0;
        int fd = open(paths[i], O_RDONLY | O_CLOEXEC);
        if (fd < 0) return; // (missing: the caller looks for the .bz2 twin, detecttrails.py:81-87)
        std::vector<char> head(4 * BLOCK);
        size_t got = 0, end = 0;
        for (;;) {
            bool eof = false;
            while (got < head.size()) {
                ssize_t r = pread(fd, head.data() + got, head.size() - got, (off_t)got);
                if (r < 0 && errno == EINTR) continue;
                if (r <= 0) { eof = true; break; }
                got += (size_t)r;
            }
            end = header_end(head.data(), got, 0);
            if (end || eof || head.size() >= 64 * BLOCK) break; // (no END card within 64 blocks: not a header)
            head.resize(head.size() * 2);
        }
        status[i] = -2;
        if (!end || end > got) { close(fd); return; } // no END card, or the file stops inside the header's last block
        if (hdr) {
            const size_t ncopy = end < (size_t)hdr_cap ? end : (size_t)hdr_cap;
            memcpy(hdr + (size_t)i * hdr_cap, head.data(), ncopy);
            hdr_len[i] = (int32_t)end; // (> hdr_cap: the copy is truncated, the caller reads the header itself)
        }
        const size_t hl = end;
        long long bitpix = 0, naxis = 0, n1 = 0, n2 = 0;
        double bscale = 1.0, bzero = 0.0;
        const bool plain = card_int(head.data(), hl, "BITPIX", &bitpix) && bitpix == -32 && card_int(head.data(), hl, "NAXIS", &naxis) &&
                           naxis == 2 && card_int(head.data(), hl, "NAXIS1", &n1) && n1 == w && card_int(head.data(), hl, "NAXIS2", &n2) &&
                           n2 == h && (!card_double(head.data(), hl, "BSCALE", &bscale) || bscale == 1.0) &&
                           (!card_double(head.data(), hl, "BZERO", &bzero) || bzero == 0.0);
        if (!plain) { status[i] = 1; close(fd); return; }
        char *slot = (char *)dst + (size_t)i * frame_bytes;
        size_t have = 0;
        if (got > end) { // data bytes that came with the header read
            have = got - end < frame_bytes ? got - end : frame_bytes;
            memcpy(slot, head.data() + end, have);
        }
        const bool ok = have == frame_bytes || read_fully(fd, slot + have, frame_bytes - have, (off_t)(end + have));
        close(fd);
        status[i] = ok ? 0 : -2;
    });
    return 0;
}

extern "C" int lfdmi_fits_read_photoobj(const char *const *paths, int n, int max_obj, float *rowc, float *colc, float *psfmag,
                                        float *petro90, int32_t *nobserve, int32_t *ndetect, int32_t *count, int threads,
                                        int32_t *status) {
    using namespace lfd_fits;
    if (!paths || !rowc || !colc || !psfmag || !petro90 || !nobserve || !ndetect || !count || !status || n < 0 || max_obj <= 0)
        return LFDMI_ERR_ARG;
    static const char *const WANT[6] = {"ROWC", "COLC", "PSFMAG", "PETROTH90", "NOBSERVE", "NDETECT"};
    parallel_for(n, threads, [&](int i) {
        status[i] = -1;
        count[i] = 0;
        int fd = open(paths[i], O_RDONLY | O_CLOEXEC);
        if (fd < 0) return;
        const off_t size = lseek(fd, 0, SEEK_END);
        std::vector<char> buf;
        if (size > 0) buf.resize((size_t)size);
        const bool ok = size > 0 && read_fully(fd, buf.data(), buf.size(), 0);
        close(fd);
        status[i] = -2;
        if (!ok) return;
        const size_t e0 = header_end(buf.data(), buf.size(), 0);
        if (!e0 || e0 > buf.size()) return; // (no END card, or the file stops inside the header's last block: e0 is rounded up to a block)
        // the primary HDU's data unit (photoObj files have none, NAXIS = 0)
        size_t off = e0;
        long long naxis = 0, bitpix = 8;
        card_int(buf.data(), e0, "NAXIS", &naxis);
        card_int(buf.data(), e0, "BITPIX", &bitpix);
        if (naxis > 0) {
            unsigned long long nb = (unsigned long long)(bitpix < 0 ? -bitpix : bitpix) / 8;
            for (int a = 1; a <= naxis; a++) {
                char key[16];
                long long na = 0;
                snprintf(key, sizeof key, "NAXIS%d", a);
                if (!card_int(buf.data(), e0, key, &na) || na < 0) return;
                if (na && nb > (unsigned long long)buf.size() / (unsigned long long)na) return; // (data unit larger than the file: also keeps the product from wrapping)
                nb *= (unsigned long long)na;
            }
            if (nb > buf.size()) return;
            off += (nb + BLOCK - 1) / BLOCK * BLOCK;
        }
        if (off >= buf.size()) return;
        const size_t e1 = header_end(buf.data(), buf.size(), off);
        if (!e1 || e1 > buf.size()) return;
        const char *th = buf.data() + off;
        const size_t tl = e1 - off;
        std::string xt;
        long long row_bytes = 0, nrows = 0, nf = 0;
        if (!card_str(th, tl, "XTENSION", &xt) || xt != "BINTABLE" || !card_int(th, tl, "NAXIS1", &row_bytes) ||
            !card_int(th, tl, "NAXIS2", &nrows) || !card_int(th, tl, "TFIELDS", &nf) || nf <= 0 || nf > 999 || row_bytes <= 0 || nrows < 0)
            return;
        // sizes from the header are checked against the file before they are used as offsets (hostile or truncated files)
        if ((unsigned long long)row_bytes > buf.size() - e1 || (nrows && (unsigned long long)nrows > (buf.size() - e1) / (unsigned long long)row_bytes)) return;
        status[i] = 1; // from here on: a well-formed table this reader may still decline
        if (nrows > max_obj) return;
        long long col_off[6], col_rep[6];
        char col_code[6];
        for (int c = 0; c < 6; c++) col_off[c] = -1;
        bool has_objc_type = false, has_type = false; // read_photoObj (removestars.py:97-104) also asks for these two columns
        long long pos = 0;
        for (int f = 1; f <= nf; f++) {
            char key[16];
            std::string form, name;
            snprintf(key, sizeof key, "TFORM%d", f);
            if (!card_str(th, tl, key, &form) || form.empty()) { status[i] = -2; return; }
            size_t j = 0;
            while (j < form.size() && form[j] >= '0' && form[j] <= '9') j++;
            const long long rep = (j && j <= 9) ? atoll(form.substr(0, j).c_str()) : (j ? -1 : 1);
            if (j >= form.size() || rep < 0 || rep > row_bytes * 8) { status[i] = -2; return; } // (a repeat count no row of this table could hold)
            const char code = form[j];
            long long width;
            switch (code) {
            case 'L': case 'B': case 'A': width = rep; break;
            case 'I': width = 2 * rep; break;
            case 'J': case 'E': width = 4 * rep; break;
            case 'K': case 'D': width = 8 * rep; break;
            case 'C': width = 8 * rep; break;
            case 'M': width = 16 * rep; break;
            case 'P': width = 8 * rep; break;
            case 'Q': width = 16 * rep; break;
            case 'X': width = (rep + 7) / 8; break;
            default: return; // unknown TFORM: declined
            }
            snprintf(key, sizeof key, "TTYPE%d", f);
            if (card_str(th, tl, key, &name)) {
                for (auto &ch : name) if (ch >= 'a' && ch <= 'z') ch = (char)(ch - 'a' + 'A');
                has_objc_type = has_objc_type || name == "OBJC_TYPE";
                has_type = has_type || name == "TYPE";
                for (int c = 0; c < 6; c++)
                    if (col_off[c] < 0 && name == WANT[c]) {
                        std::string dummy;
                        snprintf(key, sizeof key, "TSCAL%d", f);
                        const bool scaled = card(th, tl, key, &dummy);
                        snprintf(key, sizeof key, "TZERO%d", f);
                        if (scaled || card(th, tl, key, &dummy)) return; // scaled column: declined
                        col_off[c] = pos; col_rep[c] = rep; col_code[c] = code;
                    }
            }
            pos += width;
            if (pos > row_bytes) { status[i] = -2; return; }
        }
        if (pos != row_bytes) { status[i] = -2; return; }
        if (!has_objc_type || !has_type) { status[i] = -3; return; } // (KeyError in the reference and in the general reader: same outcome for every batch size)
        for (int c = 0; c < 6; c++) {
            if (col_off[c] < 0) { status[i] = -3; return; } // a wanted column is missing (KeyError in the general reader)
            if (c < 4 ? !(col_code[c] == 'E' && col_rep[c] == 5) : !((col_code[c] == 'J' || col_code[c] == 'I' || col_code[c] == 'B' || col_code[c] == 'K') && col_rep[c] == 1))
                return; // another layout than float32[5] / one integer: declined
        }
        const unsigned char *rows = (const unsigned char *)buf.data() + e1;
        float *dst5[4] = {rowc, colc, psfmag, petro90};
        bool has_nan = false, has_inf = false;
        for (long long r = 0; r < nrows; r++) {
            const unsigned char *row = rows + r * row_bytes;
            for (int c = 0; c < 4; c++) {
                float *d = dst5[c] + ((size_t)i * max_obj + (size_t)r) * 5;
                for (int k = 0; k < 5; k++) {
                    const float v = be_f32(row + col_off[c] + 4 * k);
                    d[k] = v;
                    has_nan = has_nan || v != v;
                    has_inf = has_inf || (v - v != 0.0f && v == v);
                }
            }
            const int sz4 = col_code[4] == 'J' ? 4 : (col_code[4] == 'I' ? 2 : (col_code[4] == 'K' ? 8 : 1));
            const int sz5 = col_code[5] == 'J' ? 4 : (col_code[5] == 'I' ? 2 : (col_code[5] == 'K' ? 8 : 1));
            nobserve[(size_t)i * max_obj + (size_t)r] = (int32_t)be_int(row + col_off[4], sz4);
            ndetect[(size_t)i * max_obj + (size_t)r] = (int32_t)be_int(row + col_off[5], sz5);
        }
        count[i] = (int32_t)nrows;
        status[i] = has_nan ? 2 : (has_inf ? 3 : 0); // (math.ceil(nan / inf) raises in the reference: the caller turns 2 / 3 into those errors)
    });
    return 0;
}

// Bit offsets of every 48-bit bzip2 block magic (0x314159265359) and end-of-stream magic (0x177245385090) in data[0 .. n):
// out[i] = bit offset * 2 + (1 for an end-of-stream magic), ascending; returns how many there are (only the first `cap` are
// stored).  Host code, no GPU: the first step of decoding the blocks of a .fits.bz2 frame side by side
// (lfd_amd/detecttrails/bz2blocks.py; the reference shells out to bunzip2, detecttrails.py:81-109).  A 64-bit window slides
// over the file one byte at a time; the magic can start at any of the eight bit positions of a byte.
extern "C" int64_t lfdmi_bz2_find_blocks(const uint8_t *data, uint64_t n, uint64_t *out, int64_t cap) {
    if (!data || (!out && cap > 0) || cap < 0) return LFDMI_ERR_ARG;
    const uint64_t BLK = 0x314159265359ull, EOS = 0x177245385090ull, M48 = 0xFFFFFFFFFFFFull;
    int64_t found = 0;
    // A magic whose last bit lies in byte i (k bits before the byte's end, k = 0 .. 7) covers byte i - 1 completely when k >= 1 and
    // byte i when k == 0: that byte's value is known per alignment, so one table look-up per byte says which of the eight
    // alignments are worth the full 48-bit comparison (3 % of the bytes: eight candidates of 1 / 256 each).
    uint8_t cand_prev[256] = {0}, cand_cur[256] = {0}; // bit k set: alignment k is possible given byte i - 1 / byte i
    for (uint64_t m : {BLK, EOS}) {
        cand_cur[m & 0xFF] |= 1;                                   // k = 0: byte i is the magic's last byte
        for (int k = 1; k < 8; k++) cand_prev[(m >> (8 - k)) & 0xFF] |= (uint8_t)(1u << k); // byte i - 1 = magic bits [8 - k, 16 - k) from its end
    }
    uint64_t acc = 0; // the last 8 bytes read, big-endian
    for (uint64_t i = 0; i < n; i++) {
        acc = (acc << 8) | data[i];
        if (i < 6) continue;
        unsigned ks = cand_cur[acc & 0xFF] | cand_prev[(acc >> 8) & 0xFF];
        while (ks) {
            const int k = __builtin_ctz(ks);
            ks &= ks - 1;
            const uint64_t v = (acc >> k) & M48; // 48 bits ending k bits before the end of byte i
            if (v == BLK || v == EOS) {
                const uint64_t end_bit = (i + 1) * 8 - k; // bit offset just past the magic
                if (end_bit < 48) continue;
                if (found < cap) out[found] = ((end_bit - 48) << 1) | (v == EOS ? 1u : 0u);
                found++;
            }
        }
    }
    return found;
}
