// Test harness (CPU, g++): the block decoder of lfd_amd/csrc/bz2_core.h on a whole .bz2 file -- blocks located by their magics,
// each decoded to its BWT column by the shared core, then inverse BWT + run-length expansion + CRC in plain reference code here --
// and the result written to a file for the test to compare with Python's bz2.decompress.  usage: bz2_core_check in.bz2 out.bin
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../lfd_amd/csrc/bz2_core.h"

struct HostIO {
    uint8_t len[BZ_MAX_GROUPS * BZ_MAX_ALPHA];
    int limit[BZ_MAX_GROUPS * BZ_NLEN], base[BZ_MAX_GROUPS * BZ_NLEN], min_len[BZ_MAX_GROUPS];
    uint16_t perm[BZ_MAX_GROUPS * BZ_MAX_ALPHA], fast[BZ_MAX_GROUPS * BZ_FAST_SIZE];
    uint8_t sel[BZ_MAX_SELECTORS];
    uint8_t list[256];
    std::vector<uint8_t> out;
    void mtf_begin() {}
    void mtf_add(int k, uint32_t b) { list[k] = (uint8_t)b; }
    uint32_t mtf_head() { return list[0]; }
    uint32_t mtf_front(int nn) {
        uint8_t v = list[nn];
        memmove(list + 1, list, nn);
        list[0] = v;
        return v;
    }
    void emit(uint32_t b) { out.push_back((uint8_t)b); }
    void emit_run(uint32_t b, int n) { out.insert(out.end(), n, (uint8_t)b); }
    int emitted() { return (int)out.size(); }
    void build_fast(int t, int mn) {
        for (uint32_t x = 0; x < BZ_FAST_SIZE; x++) fast[t * BZ_FAST_SIZE + x] = bz_fast_entry(*this, t, mn, x);
    }
};

static uint64_t get48(const std::vector<uint8_t> &d, uint64_t bit) {
    uint64_t v = 0;
    for (int i = 0; i < 48; i++) {
        uint64_t b = bit + i;
        v = (v << 1) | ((b >> 3) < d.size() ? (d[b >> 3] >> (7 - (b & 7))) & 1 : 0);
    }
    return v;
}

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    std::vector<uint8_t> d;
    uint8_t tmp[65536];
    size_t k;
    while ((k = fread(tmp, 1, sizeof tmp, f)) > 0) d.insert(d.end(), tmp, tmp + k);
    fclose(f);
    if (d.size() < 14 || memcmp(d.data(), "BZh", 3) || d[3] < '1' || d[3] > '9') { fprintf(stderr, "not a bzip2 stream\n"); return 3; }
    const int max_block = (d[3] - '0') * 100000;
    std::vector<uint64_t> marks; // bit offsets of block magics, then the end mark
    uint64_t eos = 0;
    for (uint64_t bit = 32; bit + 48 <= d.size() * 8; bit++) {
        uint64_t v = get48(d, bit);
        if (v == 0x314159265359ull) marks.push_back(bit);
        if (v == 0x177245385090ull) { eos = bit; break; }
    }
    if (!eos) { fprintf(stderr, "no end mark\n"); return 3; }
    marks.push_back(eos);
    std::vector<uint8_t> padded(d);
    padded.resize((d.size() + 3) / 4 * 4 + 8, 0);
    const uint32_t *words = (const uint32_t *)padded.data();
    uint32_t crc_tab[256];
    for (uint32_t i = 0; i < 256; i++) crc_tab[i] = bz_crc_table_entry(i);
    FILE *o = fopen(argv[2], "wb");
    uint32_t combined = 0;
    static HostIO io;
    for (size_t b = 0; b + 1 < marks.size(); b++) {
        io.out.clear();
        BzBlockInfo info;
        int rc = bz_decode_block(io, words, padded.size() / 4, marks[b], marks[b + 1], max_block, info);
        if (rc) { fprintf(stderr, "block %zu: status %d\n", b, rc); return 4; }
        const int n = info.nblock;
        const std::vector<uint8_t> &L = io.out;
        std::vector<uint32_t> tt(n);
        int cf[257] = {0};
        for (int i = 0; i < n; i++) cf[L[i] + 1]++;
        for (int i = 0; i < 256; i++) cf[i + 1] += cf[i];
        for (int i = 0; i < n; i++) tt[cf[L[i]]++] = ((uint32_t)i << 8) | L[i];
        std::vector<uint8_t> pre(n);
        uint32_t q = info.orig_ptr;
        for (int i = 0; i < n; i++) { uint32_t e = tt[q]; pre[i] = e & 0xff; q = e >> 8; }
        std::vector<uint8_t> out;
        for (int p = 0; p < n;) {
            uint8_t v = pre[p++];
            out.push_back(v);
            int kk = 1;
            while (p < n && kk < 4 && pre[p] == v) { out.push_back(v); p++; kk++; }
            if (kk == 4 && p < n) out.insert(out.end(), pre[p++], v);
        }
        uint32_t c = 0xffffffffu;
        for (uint8_t x : out) c = (c << 8) ^ crc_tab[(c >> 24) ^ x];
        c = ~c;
        if (c != info.crc) { fprintf(stderr, "block %zu: crc %08x stored %08x\n", b, c, info.crc); return 5; }
        combined = ((combined << 1) | (combined >> 31)) ^ c;
        fwrite(out.data(), 1, out.size(), o);
    }
    fclose(o);
    uint32_t stored = 0;
    for (int i = 0; i < 32; i++) { uint64_t bb = eos + 48 + i; stored = (stored << 1) | ((d[bb >> 3] >> (7 - (bb & 7))) & 1); }
    if (stored != combined) { fprintf(stderr, "combined crc %08x stored %08x\n", combined, stored); return 6; }
    printf("blocks %zu ok\n", marks.size() - 1);
    return 0;
}
