// bz2dev.hip -- host side of the device bzip2 decoder (kernels: k_bz2.h; C-ABI: include/lfdmi.h, lfdmi_bz2_*).
// A handle owns its stream and its device buffers; nothing here touches an lfdmi_ctx, so a loader thread can decode the next
// chunk's files while another thread drives lfdmi_detect_batch on the same GPU.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/lfdmi.h"
#include "k_bz2.h"

struct lfdmi_bz2 {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    // device buffers, grown on demand
    uint32_t *comp = nullptr; size_t comp_words = 0;
    uint64_t *word_off = nullptr, *nbytes = nullptr; int *nfound = nullptr; u64 *marks = nullptr; size_t files_cap = 0;
    int *file_first = nullptr, *file_status = nullptr; u64 *out_len = nullptr;
    BzBlockDesc *desc = nullptr; BzBlockInfo *info = nullptr; uint8_t *Lbuf = nullptr, *selbuf = nullptr, *segbuf = nullptr; uint32_t *tt = nullptr;
    int4 *meta = nullptr, *tile_fn = nullptr; int2 *tile_in = nullptr; uint32_t *blk_crc = nullptr;
    int *blk_size = nullptr; u64 *blk_off = nullptr; size_t blocks_cap = 0;
    uint8_t *out = nullptr; size_t out_bytes = 0;
    uint8_t *heads = nullptr; size_t heads_bytes = 0;
    uint8_t *frames[2] = {nullptr, nullptr}; size_t frames_bytes[2] = {0, 0}; // what the caller gathers decoded data units into (lfdmi_bz2_frames)
    // the last batch
    int n_files = 0; uint64_t out_cap = 0;
    std::vector<uint64_t> h_out_len;
    std::vector<int> h_status;
    BzCrcPow pows;
    // timings of the last batch (ms): upload + magics, Huffman, sort, walk, run-length + output
    float ms[5] = {0, 0, 0, 0, 0};
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
};

static int bfail(lfdmi_bz2 *z, int code, const std::string &msg) {
    if (z) z->err = msg;
    return code;
}
#define BCHK(expr)                                                                              \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return bfail(z, LFDMI_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

template <class T> static hipError_t regrow(T *&p, size_t &have, size_t want) { // (contents are not kept)
    if (want <= have) return hipSuccess;
    if (p) { (void)hipFree(p); p = nullptr; have = 0; }
    hipError_t e = hipMalloc((void **)&p, want * sizeof(T));
    if (e == hipSuccess) have = want;
    return e;
}
template <class T> static hipError_t alloc_n(T *&p, size_t n) {
    if (p) { (void)hipFree(p); p = nullptr; }
    return hipMalloc((void **)&p, n * sizeof(T));
}

static uint32_t gf_mul_host(uint32_t a, uint32_t b) {
    uint32_t r = 0;
    for (int i = 31; i >= 0; i--) {
        r = (r << 1) ^ ((r & 0x80000000u) ? 0x04c11db7u : 0u);
        if ((b >> i) & 1u) r ^= a;
    }
    return r;
}

static int reserve_blocks(lfdmi_bz2 *z, size_t B) { // tables and scratch for B blocks in flight (contents are not kept)
    if (B <= z->blocks_cap) return 0;
    const size_t cap = B + B / 8 + 16;
    BCHK(alloc_n(z->desc, cap)); BCHK(alloc_n(z->info, cap)); BCHK(alloc_n(z->Lbuf, cap * BZ_LSTRIDE));
    BCHK(alloc_n(z->selbuf, cap * BZ_SEL_STRIDE)); BCHK(alloc_n(z->segbuf, cap * BZ_MAX_SPLIT * BZ_SEG_CAP)); BCHK(alloc_n(z->tt, cap * BZ_TSTRIDE)); BCHK(alloc_n(z->meta, cap * BZ_MAX_TILES * 1024));
    BCHK(alloc_n(z->tile_fn, cap * BZ_MAX_TILES)); BCHK(alloc_n(z->tile_in, cap * BZ_MAX_TILES)); BCHK(alloc_n(z->blk_crc, cap));
    BCHK(alloc_n(z->blk_size, cap)); BCHK(alloc_n(z->blk_off, cap));
    z->blocks_cap = cap;
    return 0;
}

// Allocates ahead of the first lfdmi_bz2_decode_batch what a batch of n_files files with n_blocks blocks in all, out_cap bytes of
// output each, will need (tens of GB for a chunk of frames: half a second of hipMalloc that can run beside the first reads).
extern "C" int lfdmi_bz2_reserve(lfdmi_bz2 *z, int n_files, int64_t n_blocks, uint64_t out_cap, uint64_t compressed_bytes) {
    if (!z || n_files <= 0 || n_blocks < 0) return bfail(z, LFDMI_ERR_ARG, "lfdmi_bz2_reserve: bad argument");
    BCHK(hipSetDevice(z->device));
    size_t max_blocks = 4096;
    if (const char *e = getenv("LFDMI_BZ2_MAX_BLOCKS")) max_blocks = (size_t)std::max(1, atoi(e));
    { int rc_ = reserve_blocks(z, std::min((size_t)n_blocks, max_blocks)); if (rc_) return rc_; }
    out_cap = (out_cap + 255) & ~(uint64_t)255;
    BCHK(regrow(z->out, z->out_bytes, (size_t)n_files * out_cap + 256));
    if (compressed_bytes) BCHK(regrow(z->comp, z->comp_words, (size_t)(compressed_bytes / 4) + (size_t)n_files * 128 + 64));
    return 0;
}

extern "C" int lfdmi_bz2_create(int device, lfdmi_bz2 **out) {
    if (!out) return LFDMI_ERR_ARG;
    *out = nullptr;
    lfdmi_bz2 *z = new lfdmi_bz2();
    z->device = device;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&z->stream, hipStreamNonBlocking);
    for (int i = 0; i < 6 && e == hipSuccess; i++) e = hipEventCreate(&z->ev[i]);
    if (e != hipSuccess) { delete z; return LFDMI_ERR_HIP; }
    uint32_t p = 0x100u; // x^8
    for (int k = 0; k < 24; k++) { z->pows.pw[k] = p; p = gf_mul_host(p, p); }
    *out = z;
    return 0;
}

extern "C" void lfdmi_bz2_destroy(lfdmi_bz2 *z) {
    if (!z) return;
    (void)hipSetDevice(z->device);
    if (z->stream) (void)hipStreamSynchronize(z->stream);
    void *ps[] = {z->comp, z->word_off, z->nbytes, z->nfound, z->marks, z->file_first, z->file_status, z->out_len, z->desc, z->info,
                  z->Lbuf, z->selbuf, z->segbuf, z->tt, z->meta, z->tile_fn, z->tile_in, z->blk_crc, z->blk_size, z->blk_off, z->out, z->heads, z->frames[0], z->frames[1]};
    for (void *p : ps) if (p) (void)hipFree(p);
    for (int i = 0; i < 6; i++) if (z->ev[i]) (void)hipEventDestroy(z->ev[i]);
    if (z->stream) (void)hipStreamDestroy(z->stream);
    delete z;
}

extern "C" const char *lfdmi_bz2_last_error(lfdmi_bz2 *z) { return z ? z->err.c_str() : "null handle"; }

extern "C" int lfdmi_bz2_timings(lfdmi_bz2 *z, float *ms5) {
    if (!z || !ms5) return LFDMI_ERR_ARG;
    for (int i = 0; i < 5; i++) ms5[i] = z->ms[i];
    return 0;
}

static uint32_t bits32(const uint8_t *d, uint64_t n, uint64_t bit) {
    uint32_t v = 0;
    for (int i = 0; i < 32; i++) {
        const uint64_t b = bit + i;
        v = (v << 1) | ((b >> 3) < n ? (uint32_t)((d[b >> 3] >> (7 - (b & 7))) & 1) : 0u);
    }
    return v;
}

extern "C" int lfdmi_bz2_decode_batch(lfdmi_bz2 *z, const void *src, const uint64_t *src_off, const uint64_t *src_len, int n,
                                      uint64_t out_cap, void *head, uint64_t head_bytes, uint64_t *out_len, int32_t *status) {
    if (!z || !src || !src_off || !src_len || n <= 0 || !out_len || !status || out_cap == 0 || (head_bytes && !head))
        return bfail(z, LFDMI_ERR_ARG, "lfdmi_bz2_decode_batch: bad argument");
    BCHK(hipSetDevice(z->device));
    const uint8_t *S = (const uint8_t *)src;
    out_cap = (out_cap + 255) & ~(uint64_t)255;
    z->n_files = 0;
    // ---- the compressed bytes, every file on a 256-byte boundary and followed by zero words
    std::vector<uint64_t> woff(n), nby(n);
    size_t words = 0;
    uint64_t longest = 0;
    for (int i = 0; i < n; i++) {
        woff[i] = words;
        nby[i] = src_len[i];
        longest = std::max(longest, src_len[i]);
        words += ((src_len[i] + 3) / 4 + 64 + 63) & ~(size_t)63;
    }
    BCHK(regrow(z->comp, z->comp_words, words + 64));
    if ((size_t)n > z->files_cap) {
        BCHK(alloc_n(z->word_off, (size_t)n)); BCHK(alloc_n(z->nbytes, (size_t)n)); BCHK(alloc_n(z->nfound, (size_t)n));
        BCHK(alloc_n(z->marks, (size_t)n * BZ_MARK_CAP)); BCHK(alloc_n(z->file_first, (size_t)n + 1));
        BCHK(alloc_n(z->file_status, (size_t)n)); BCHK(alloc_n(z->out_len, (size_t)n));
        z->files_cap = (size_t)n;
    }
    BCHK(hipEventRecord(z->ev[0], z->stream));
    BCHK(hipMemsetAsync(z->comp, 0, (words + 64) * 4, z->stream));
    for (int i = 0; i < n; i++)
        if (src_len[i]) BCHK(hipMemcpyAsync(z->comp + woff[i], S + src_off[i], src_len[i], hipMemcpyHostToDevice, z->stream));
    BCHK(hipMemcpyAsync(z->word_off, woff.data(), n * sizeof(uint64_t), hipMemcpyHostToDevice, z->stream));
    BCHK(hipMemcpyAsync(z->nbytes, nby.data(), n * sizeof(uint64_t), hipMemcpyHostToDevice, z->stream));
    BCHK(hipMemsetAsync(z->nfound, 0, n * sizeof(int), z->stream));
    {
        const unsigned gx = (unsigned)((longest / 8 + 256) / 256);
        k_bz2_magics<<<dim3(gx, n), 256, 0, z->stream>>>(z->comp, z->word_off, z->nbytes, z->nfound, z->marks);
        BCHK(hipGetLastError());
    }
    std::vector<int> nfound(n);
    std::vector<u64> marks((size_t)n * BZ_MARK_CAP);
    BCHK(hipMemcpyAsync(nfound.data(), z->nfound, n * sizeof(int), hipMemcpyDeviceToHost, z->stream));
    BCHK(hipMemcpyAsync(marks.data(), z->marks, marks.size() * sizeof(u64), hipMemcpyDeviceToHost, z->stream));
    BCHK(hipStreamSynchronize(z->stream));
    // ---- the blocks of every file that is one plain stream
    std::vector<BzBlockDesc> desc;
    std::vector<int> first(n + 1), fstat(n, BZ_OK);
    struct Stream { int file, b0, b1; uint32_t crc; }; // a stream's blocks [b0, b1) and its stored CRC
    std::vector<Stream> streams;
    for (int i = 0; i < n; i++) {
        first[i] = (int)desc.size();
        const uint8_t *d = S + src_off[i];
        const uint64_t len = src_len[i];
        if (len < 14 || nfound[i] < 1 || nfound[i] > BZ_MARK_CAP) { fstat[i] = BZ_E_STREAM; continue; }
        u64 *m = marks.data() + (size_t)i * BZ_MARK_CAP;
        const int nm = nfound[i];
        std::sort(m, m + nm);
        // One or more streams, each "BZh" + level, blocks, end mark + CRC, padded to a byte (bzip2 -c a b, pbzip2).  Every magic
        // the device found must be where this walk expects one: anything else (bytes that are not bzip2, a magic-like pattern inside
        // compressed data) declines the file.
        const size_t desc0 = desc.size(), streams0 = streams.size();
        uint64_t byte = 0;
        int k = 0;
        bool ok = true;
        while (ok && byte < len) {
            ok = byte + 14 <= len && d[byte] == 'B' && d[byte + 1] == 'Z' && d[byte + 2] == 'h' && d[byte + 3] >= '1' && d[byte + 3] <= '9' &&
                 k < nm && (m[k] >> 1) == byte * 8 + 32;
            if (!ok) break;
            const int max_block = (d[byte + 3] - '0') * 100000;
            Stream st;
            st.file = i;
            st.b0 = (int)desc.size();
            while (k < nm && !(m[k] & 1)) { // blocks up to the end mark
                if (k + 1 >= nm) { ok = false; break; }
                BzBlockDesc bd;
                bd.start_bit = m[k] >> 1;
                bd.end_bit = m[k + 1] >> 1;
                bd.word_off = woff[i];
                bd.nwords = (len + 3) / 4 + 2;
                bd.file = i;
                bd.max_block = max_block;
                desc.push_back(bd);
                k++;
            }
            if (!ok || k >= nm) { ok = false; break; }
            const uint64_t eos = m[k] >> 1;
            k++;
            if ((eos + 80 + 7) / 8 > len) { ok = false; break; }
            st.b1 = (int)desc.size();
            st.crc = bits32(d, len, eos + 48);
            streams.push_back(st);
            byte = (eos + 80 + 7) / 8;
        }
        if (ok) ok = byte == len && k == nm;
        if (!ok) {
            fstat[i] = BZ_E_STREAM;
            desc.resize(desc0);
            streams.resize(streams0);
        }
    }
    first[n] = (int)desc.size();
    // ---- groups of whole files, each with at most max_blocks blocks in flight (8.6 MB of tables and scratch per block): the
    // files of a chunk of SDSS frames are one group; what is larger (more files, level-1 files of 127 blocks) takes several passes
    size_t max_blocks = 4096;
    if (const char *e = getenv("LFDMI_BZ2_MAX_BLOCKS")) max_blocks = (size_t)std::max(1, atoi(e));
    std::vector<int> gfirst{0}; // first file of every group, and n
    {
        size_t inb = 0;
        for (int i = 0; i < n; i++) {
            const size_t nb = (size_t)(first[i + 1] - first[i]);
            if (inb && inb + nb > max_blocks) { gfirst.push_back(i); inb = 0; }
            inb += nb;
        }
        gfirst.push_back(n);
    }
    size_t Bmax = 0;
    for (size_t g = 0; g + 1 < gfirst.size(); g++) Bmax = std::max(Bmax, (size_t)(first[gfirst[g + 1]] - first[gfirst[g]]));
    { int rc_ = reserve_blocks(z, Bmax); if (rc_) return rc_; }
    BCHK(regrow(z->out, z->out_bytes, (size_t)n * out_cap + 256));
    if (head_bytes) BCHK(regrow(z->heads, z->heads_bytes, (size_t)n * head_bytes));
    BCHK(hipMemcpyAsync(z->file_status, fstat.data(), n * sizeof(int), hipMemcpyHostToDevice, z->stream));
    std::vector<BzBlockInfo> info(desc.size());
    float ms_acc[5] = {0, 0, 0, 0, 0};
    std::vector<std::vector<int>> lfirsts;
    lfirsts.reserve(gfirst.size());
    for (size_t g = 0; g + 1 < gfirst.size(); g++) {
        const int f0 = gfirst[g], nf = gfirst[g + 1] - f0, b0 = first[f0];
        const size_t B = (size_t)(first[f0 + nf] - b0);
        lfirsts.emplace_back(nf + 1); // the group's files -> its blocks, counted from the group's first (kept until the stream is idle)
        std::vector<int> &lfirst = lfirsts.back();
        for (int i = 0; i <= nf; i++) lfirst[i] = first[f0 + i] - b0;
        BCHK(hipMemcpyAsync(z->file_first, lfirst.data(), (nf + 1) * sizeof(int), hipMemcpyHostToDevice, z->stream));
        if (B) {
            BCHK(hipMemcpyAsync(z->desc, desc.data() + b0, B * sizeof(BzBlockDesc), hipMemcpyHostToDevice, z->stream));
            BCHK(hipEventRecord(z->ev[1], z->stream));
            k_bz2_huff<<<(unsigned)B, 64, 0, z->stream>>>(z->comp, z->desc, z->info, z->Lbuf, z->selbuf, (int)B);
            BCHK(hipGetLastError());
            BCHK(hipEventRecord(z->ev[2], z->stream));
            k_bz2_sort<<<(unsigned)B, 1024, 0, z->stream>>>(z->info, z->Lbuf, z->tt);
            BCHK(hipGetLastError());
            BCHK(hipEventRecord(z->ev[3], z->stream));
            k_bz2_walk<<<(unsigned)B, 1024, 0, z->stream>>>(z->info, z->tt, z->Lbuf, z->segbuf);
            BCHK(hipGetLastError());
            BCHK(hipEventRecord(z->ev[4], z->stream));
            k_bz2_rle_tiles<<<dim3(BZ_MAX_TILES, (unsigned)B), 1024, 0, z->stream>>>(z->info, z->Lbuf, z->meta, z->tile_fn);
            BCHK(hipGetLastError());
            k_bz2_rle_blocks<<<(unsigned)((B + 63) / 64), 64, 0, z->stream>>>(z->info, z->tile_fn, z->tile_in, z->blk_size, (int)B);
            BCHK(hipGetLastError());
            BCHK(hipMemsetAsync(z->blk_crc, 0, B * sizeof(uint32_t), z->stream));
        } else {
            for (int k = 1; k <= 4; k++) BCHK(hipEventRecord(z->ev[k], z->stream));
        }
        k_bz2_offsets<<<(nf + 63) / 64, 64, 0, z->stream>>>(z->info, z->blk_size, z->file_first, nf, out_cap, z->blk_off, z->out_len + f0, z->file_status + f0);
        BCHK(hipGetLastError());
        if (B) {
            k_bz2_expand<<<dim3(BZ_MAX_TILES, (unsigned)B), 1024, 0, z->stream>>>(z->info, z->desc, z->Lbuf, z->meta, z->tile_fn, z->tile_in, z->blk_size,
                                                                                 z->blk_off, z->out, out_cap, z->file_status, z->blk_crc, z->pows);
            BCHK(hipGetLastError());
            k_bz2_crc_check<<<(unsigned)((B + 63) / 64), 64, 0, z->stream>>>(z->info, z->desc, z->blk_crc, z->file_status, (int)B);
            BCHK(hipGetLastError());
            BCHK(hipMemcpyAsync(info.data() + b0, z->info, B * sizeof(BzBlockInfo), hipMemcpyDeviceToHost, z->stream));
        }
        BCHK(hipEventRecord(z->ev[5], z->stream));
        BCHK(hipStreamSynchronize(z->stream)); // (the next group reuses the tables)
        for (int k = 0; k < 5; k++) {
            float t = 0;
            (void)hipEventElapsedTime(&t, z->ev[k], z->ev[k + 1]);
            if (k > 0 || g == 0) ms_acc[k] += t; // (the upload and the magic search happen once, before the first group)
        }
    }
    for (int k = 0; k < 5; k++) z->ms[k] = ms_acc[k];
    if (head_bytes) {
        k_bz2_heads<<<dim3((unsigned)((head_bytes + 255) / 256), n), 256, 0, z->stream>>>(z->out, out_cap, z->out_len, z->heads, head_bytes);
        BCHK(hipGetLastError());
        BCHK(hipMemcpyAsync(head, z->heads, (size_t)n * head_bytes, hipMemcpyDeviceToHost, z->stream));
    }
    z->h_out_len.assign(n, 0);
    z->h_status.assign(n, 0);
    BCHK(hipMemcpyAsync(z->h_out_len.data(), z->out_len, n * sizeof(u64), hipMemcpyDeviceToHost, z->stream));
    BCHK(hipMemcpyAsync(z->h_status.data(), z->file_status, n * sizeof(int), hipMemcpyDeviceToHost, z->stream));
    BCHK(hipStreamSynchronize(z->stream));
    for (const Stream &st : streams) { // every stream's own CRC: its blocks' CRCs (each one checked on the device) rotated together
        if (z->h_status[st.file] != BZ_OK) continue;
        uint32_t c = 0;
        for (int b = st.b0; b < st.b1; b++) c = ((c << 1) | (c >> 31)) ^ info[b].crc;
        if (c != st.crc) z->h_status[st.file] = BZ_E_CRC;
    }
    for (int i = 0; i < n; i++) {
        status[i] = z->h_status[i];
        out_len[i] = z->h_status[i] == BZ_OK ? z->h_out_len[i] : 0;
    }
    z->n_files = n;
    z->out_cap = out_cap;
    return 0;
}

extern "C" int lfdmi_bz2_fetch(lfdmi_bz2 *z, int i, uint64_t off, uint64_t nbytes, void *dst, int loc) {
    if (!z || !dst || i < 0 || i >= z->n_files) return bfail(z, LFDMI_ERR_ARG, "lfdmi_bz2_fetch: bad argument");
    if (z->h_status[i] != BZ_OK || off + nbytes > z->h_out_len[i]) return bfail(z, LFDMI_ERR_ARG, "lfdmi_bz2_fetch: outside the decoded file");
    BCHK(hipSetDevice(z->device));
    BCHK(hipMemcpyAsync(dst, z->out + (size_t)i * z->out_cap + off, nbytes, loc == LFDMI_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost,
                        z->stream));
    BCHK(hipStreamSynchronize(z->stream));
    return 0;
}

// several ranges in one go: n copies queued, one wait
extern "C" int lfdmi_bz2_fetch_many(lfdmi_bz2 *z, int n, const int32_t *file, const uint64_t *off, const uint64_t *nbytes, void *const *dst, int loc) {
    if (!z || n < 0 || (n && (!file || !off || !nbytes || !dst))) return bfail(z, LFDMI_ERR_ARG, "lfdmi_bz2_fetch_many: bad argument");
    BCHK(hipSetDevice(z->device));
    for (int k = 0; k < n; k++) {
        const int i = file[k];
        if (i < 0 || i >= z->n_files || !dst[k] || z->h_status[i] != BZ_OK || off[k] + nbytes[k] > z->h_out_len[i])
            return bfail(z, LFDMI_ERR_ARG, "lfdmi_bz2_fetch_many: outside a decoded file");
        BCHK(hipMemcpyAsync(dst[k], z->out + (size_t)i * z->out_cap + off[k], nbytes[k],
                            loc == LFDMI_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, z->stream));
    }
    BCHK(hipStreamSynchronize(z->stream));
    return 0;
}

// device memory of the handle's own for the caller to gather decoded ranges into (lfdmi_bz2_fetch_many with LFDMI_DEVICE) and hand to
// lfdmi_detect_batch_raw(..., LFDMI_F32_BE, ..., LFDMI_DEVICE): two buffers, so that one chunk can be decoded while the previous
// one is being processed.  A buffer keeps its address until a larger one is asked for under the same index.
extern "C" int lfdmi_bz2_frames(lfdmi_bz2 *z, int which, uint64_t bytes, void **dev) {
    if (!z || !dev || which < 0 || which > 1 || bytes == 0) return bfail(z, LFDMI_ERR_ARG, "lfdmi_bz2_frames: bad argument");
    BCHK(hipSetDevice(z->device));
    BCHK(regrow(z->frames[which], z->frames_bytes[which], (size_t)bytes));
    *dev = z->frames[which];
    return 0;
}
