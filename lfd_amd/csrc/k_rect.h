// k_rect.h -- contour keys -> per-row extremes -> convex hull -> rotating calipers
// (cv2.minAreaRect) -> lfd's side / elongation filter -> cv2.boxPoints -> int32 truncation ->
// cv2.fillPoly into a bit-row box image.  Reference: processfield.py:248-261.
//
// A "key" is one contour of cv2.findContours(RETR_LIST): an 8-connected edge component
// (its outer border) or a hole (its hole border), see k_ccl.h.  Only the strictly convex hull
// of a contour matters downstream, and a hull vertex is always the left-most or right-most
// contour point of its row, so each key keeps (xmin, xmax) per row ("slots") and the hull is
// two monotone chains over those.  Float arithmetic restates OpenCV's rotcalipers.cpp in
// float32 evaluation order (compile with -ffp-contract=off).
#pragma once
#include "common.h"

#define KEY_HOLE_BIT (1 << 30)
#define SMALL_KEY_ROWS 16 // keys up to this many rows: one thread each (k_rects)
#define BIG_KEY_ROWS 64   // up to this many rows: one wave each with a 64-row LDS footprint; taller: one wave, full-height LDS

// Run tables of one frame slot (compact ids, see k_ccl.h)
struct RunTabs {
    const u64 *cand, *edge;          // bit rows (edge runs are candidate runs: fg ids come from cand)
    const int *scanf, *scanb;        // per-word exclusive run counts
    int *Lf, *YMf, *SBf, *ROWf;      // edge-run labels, last row / slot base per root, row per run
    const int *FLf;                  // root of an edge component (candidate component holding a strong pixel)
    int *Lb, *YMb, *FLb, *SBb, *PAb, *ROWb; // background runs: label, last row, outside flag, slot base, parent, row
    int run_cap;
};

__global__ void __launch_bounds__(256)
k_keys(RunTabs t, int4 *keys, int *bigkeys, int *medkeys, int2 *rowext, int *counters, int h, int w, int key_cap, int slot_cap,
       const int *wlist_fg, const int *wlist_bg, const int *active) {
    int g = blockIdx.y;
    if (slot_off(active, counters, g)) return;
    int wq = LFD_WQ(w);
    size_t fo = (size_t)g * h * wq, ro = (size_t)g * t.run_cap;
    int *cnt = counters + g * C_COUNT;
    int4 *kg = keys + (size_t)g * key_cap;
    int2 *re = rowext + (size_t)g * slot_cap;
    // edge components come from the candidate-word list, holes from the background-word list
    for (int val = 1; val >= 0; val--) {
        const int nwork = cnt[val ? C_NFGW : C_NBGW];
        const int *wl = (val ? wlist_fg : wlist_bg) + fo;
        for (int it = blockIdx.x * 256 + threadIdx.x; it < nwork; it += gridDim.x * 256) {
            int idx = wl[it];
            int y = idx / wq, q = idx - y * wq;
            const u64 *row = t.edge + fo + (size_t)y * wq;
            u64 s = start_bits(row, q, val, w);
            // edge runs are numbered among the candidate runs; background runs have their own scan
            int bid = val ? 0 : t.scanb[fo + idx];
            for (; s; bid++) {
                int b = __ffsll((long long)s) - 1;
                s &= s - 1;
                int x = (q << 6) + b;
                int id, ymin, extent, parent = -1;
                if (val) {
                    id = run_id(t.scanf + fo, t.cand + fo, y, x, 1, wq, w);
                    if (id < 0 || id >= t.run_cap || t.Lf[ro + id] != id) continue;
                    ymin = y;
                    extent = t.YMf[ro + id] - y + 1;
                } else {
                    id = bid;
                    if (id >= t.run_cap || t.Lb[ro + id] != id || t.FLb[ro + id]) continue;
                    // hole: border rows run from the row above its first pixel to the row below its last
                    ymin = y - 1;
                    extent = t.YMb[ro + id] - y + 3;
                    parent = t.Lf[ro + run_id(t.scanf + fo, t.cand + fo, y - 1, x, 1, wq, w)];
                }
                int base = atomicAdd(&cnt[C_NSLOTS], extent);
                int ki = atomicAdd(&cnt[C_NKEYS], 1);
                if (base + extent > slot_cap || ki >= key_cap) {
                    cnt[C_OVERFLOW] = 1;
                    if (val) t.SBf[ro + id] = -1; else t.SBb[ro + id] = -1;
                    continue;
                }
                if (val) t.SBf[ro + id] = base;
                else { t.SBb[ro + id] = base; t.PAb[ro + id] = parent; }
                kg[ki] = make_int4(id, extent | (val ? 0 : KEY_HOLE_BIT), ymin, base);
                if (extent > BIG_KEY_ROWS) bigkeys[(size_t)g * key_cap + atomicAdd(&cnt[C_NBIG], 1)] = ki;
                else if (extent > SMALL_KEY_ROWS) medkeys[(size_t)g * key_cap + atomicAdd(&cnt[C_NMED], 1)] = ki;
                for (int r = 0; r < extent; r++) re[base + r] = make_int2(0x7fffffff, -1);
            }
        }
    }
}

__device__ __forceinline__ void slot_update(int2 *re, int slot, int xa, int xb) {
    atomicMin(&re[slot].x, xa);
    atomicMax(&re[slot].y, xb);
}

// Every edge run contributes to the outer-border key of its component, and to the hole-border
// key of every hole it is 4-adjacent to (if its component is that hole's surrounding one).
__global__ void __launch_bounds__(256)
k_extremes(RunTabs t, int2 *rowext, int h, int w, int slot_cap, const int *wlist, const int *counters,
           const int *active) {
    int g = blockIdx.y;
    if (slot_off(active, counters, g)) return;
    int wq = LFD_WQ(w);
    size_t fo = (size_t)g * h * wq, ro = (size_t)g * t.run_cap;
    const int nwork = counters[g * C_COUNT + C_NFGW];
    const int *wl = wlist + fo;
    const u64 *eb = t.edge + fo;
    const int *sb = t.scanb + fo;
    const int *Lbg = t.Lb + ro, *FLbg = t.FLb + ro, *SBbg = t.SBb + ro, *PAbg = t.PAb + ro, *ROWbg = t.ROWb + ro;
    int2 *re = rowext + (size_t)g * slot_cap;
    for (int it = blockIdx.x * 256 + threadIdx.x; it < nwork; it += gridDim.x * 256) {
    int idx = wl[it];
    int y = idx / wq, q = idx - y * wq;
    const u64 *row = eb + (size_t)y * wq;
    u64 s = start_bits(row, q, 1, w);
    while (s) {
        int b = __ffsll((long long)s) - 1;
        s &= s - 1;
        int xs = (q << 6) + b;
        int xe = run_end(row, xs, 1, w);
        int fid = run_id(t.scanf + fo, t.cand + fo, y, xs, 1, wq, w);
        if (fid < 0 || fid >= t.run_cap) continue;
        int A = t.Lf[ro + fid];
        int baseA = t.SBf[ro + A];
        if (baseA >= 0) slot_update(re, baseA + (y - t.ROWf[ro + A]), xs, xe);
        // same-row neighbours
        if (xs > 0) {
            int B = Lbg[run_id(sb, eb, y, xs - 1, 0, wq, w)];
            if (!FLbg[B] && PAbg[B] == A && SBbg[B] >= 0) slot_update(re, SBbg[B] + (y - (ROWbg[B] - 1)), xs, xs);
        }
        if (xe < w - 1) {
            int B = Lbg[run_id(sb, eb, y, xe + 1, 0, wq, w)];
            if (!FLbg[B] && PAbg[B] == A && SBbg[B] >= 0) slot_update(re, SBbg[B] + (y - (ROWbg[B] - 1)), xe, xe);
        }
        // rows above and below: 0-runs overlapping [xs, xe]
        for (int dy = -1; dy <= 1; dy += 2) {
            int yy = y + dy;
            if (yy < 0 || yy >= h) continue;
            const u64 *orow = row + dy * wq;
            int x = xs;
            while (x <= xe) {
                if (get_bit(orow, x)) { x = run_end(orow, x, 1, w) + 1; continue; }
                int ge = run_end(orow, x, 0, w);
                if (ge > xe) ge = xe;
                int B = Lbg[run_id(sb, eb, yy, x, 0, wq, w)];
                if (!FLbg[B] && PAbg[B] == A && SBbg[B] >= 0)
                    slot_update(re, SBbg[B] + (y - (ROWbg[B] - 1)), x, ge);
                x = ge + 1;
            }
        }
    }
    }
}

// cv2.RETR_EXTERNAL (processfield.py:226-230 lets any retrieval mode through): only the outer borders of components that
// no other component encloses.  Suzuki-Abe decides that while scanning ("the last border pixel met on this row is
// positive"); the same set is: edge components whose raster-first pixel has the OUTSIDE
// background (the 4-connected 0-component that touches the frame) as its left neighbour, or sits in column 0
// (tests/test_contour_equivalence.py checks the two rules against each other).  A key's first row slot holds that
// pixel's column, so this runs after the row extremes: hole keys and enclosed components get extent 0 and the
// rectangle kernels skip them.
__global__ void __launch_bounds__(256)
k_filter_external(RunTabs t, int4 *keys, const int2 *rowext, const int *counters, int h, int w, int key_cap, int slot_cap,
                  const int *active) {
    int g = blockIdx.y;
    if (slot_off(active, counters, g)) return;
    const int wq = LFD_WQ(w);
    const size_t fo = (size_t)g * h * wq, ro = (size_t)g * t.run_cap;
    int nkeys = min(counters[g * C_COUNT + C_NKEYS], key_cap);
    int4 *kg = keys + (size_t)g * key_cap;
    const int2 *re = rowext + (size_t)g * slot_cap;
    for (int ki = blockIdx.x * 256 + threadIdx.x; ki < nkeys; ki += gridDim.x * 256) {
        int4 key = kg[ki];
        bool keep = !(key.y & KEY_HOLE_BIT);
        if (keep) {
            int y0 = key.z, x0 = re[key.w].x;
            if (x0 > 0 && x0 < w) {
                int bid = run_id(t.scanb + fo, t.edge + fo, y0, x0 - 1, 0, wq, w);
                keep = bid >= 0 && bid < t.run_cap && t.FLb[ro + t.Lb[ro + bid]] != 0;
            }
        }
        if (!keep) kg[ki].y = key.y & KEY_HOLE_BIT; // extent 0
    }
}

// ---- minAreaRect on a hull given through an accessor -------------------------------------
struct HullView {
    const int2 *c1; // chain 1 (left side, rows increasing)
    const int2 *c2; // chain 2 (right side, rows decreasing), already trimmed
    int n1, n, s0;  // n = total vertices, s0 = rotation so that vertex 0 is (min x, then min y)
    __device__ __forceinline__ int2 raw(int i) const { return i < n1 ? c1[i] : c2[i - n1]; }
    __device__ __forceinline__ int2 at(int i) const {
        int k = i + s0;
        if (k >= n) k -= n;
        return raw(k);
    }
    __device__ __forceinline__ float px(int i) const { return (float)at(i).x; }
    __device__ __forceinline__ float py(int i) const { return (float)at(i).y; }
};

// the same over lane-interleaved chains (k_rects: a lane per key, element k of a lane's chain at k * 64)
struct LaneHullView {
    const int2 *c1, *c2;
    int n1, n, s0;
    __device__ __forceinline__ int2 raw(int i) const { return i < n1 ? c1[i * 64] : c2[(i - n1) * 64]; }
    __device__ __forceinline__ int2 at(int i) const {
        int k = i + s0;
        if (k >= n) k -= n;
        return raw(k);
    }
    __device__ __forceinline__ float px(int i) const { return (float)at(i).x; }
    __device__ __forceinline__ float py(int i) const { return (float)at(i).y; }
};

template <class HV>
__device__ __forceinline__ void hv_vect(const HV &hv, int i, float *vx, float *vy, float *inv) {
    int j = (i + 1 < hv.n) ? i + 1 : 0;
    int2 a = hv.at(i), b = hv.at(j);
    double dx = (double)__fsub_rn((float)b.x, (float)a.x);
    double dy = (double)__fsub_rn((float)b.y, (float)a.y);
    *vx = (float)dx;
    *vy = (float)dy;
    *inv = (float)(1. / sqrt(dx * dx + dy * dy));
}

// rotcalipers.cpp CALIPERS_MINAREARECT, n > 2.  out = corner, vec1, vec2 (6 floats).
// (core: from the four extreme vertices on; rotating_calipers_dev finds them with the sequential scan of the source, the
// wave-per-key kernel with a wave-wide one that resolves ties the same way: lowest index)
template <class HV>
__device__ void rotating_calipers_core(const HV &hv, int left, int bottom, int right, int top, float *out) {
    int n = hv.n;
    float minarea = 3.402823466e+38f;
    int buf_i0 = 0, buf_i5 = 0;
    float buf1 = 0, buf2 = 0, buf3 = 0, buf4 = 0;
    int seq[4];
    float orientation = 0, base_a, base_b = 0;
    {
        float vx, vy, il;
        hv_vect(hv, n - 1, &vx, &vy, &il);
        double ax = vx, ay = vy;
        for (int i = 0; i < n; i++) {
            hv_vect(hv, i, &vx, &vy, &il);
            double bx = vx, by = vy;
            double convexity = ax * by - ay * bx;
            if (convexity != 0) { orientation = (convexity > 0) ? 1.f : (-1.f); break; }
            ax = bx; ay = by;
        }
    }
    base_a = orientation;
    seq[0] = bottom; seq[1] = right; seq[2] = top; seq[3] = left;
    float vx[4], vy[4], il[4];
    for (int i = 0; i < 4; i++) hv_vect(hv, seq[i], &vx[i], &vy[i], &il[i]);
    for (int k = 0; k < n; k++) {
        float dp[4];
        dp[0] = __fadd_rn(__fmul_rn(+base_a, vx[0]), __fmul_rn(base_b, vy[0]));
        dp[1] = __fadd_rn(__fmul_rn(-base_b, vx[1]), __fmul_rn(base_a, vy[1]));
        dp[2] = __fsub_rn(__fmul_rn(-base_a, vx[2]), __fmul_rn(base_b, vy[2]));
        dp[3] = __fsub_rn(__fmul_rn(+base_b, vx[3]), __fmul_rn(base_a, vy[3]));
        float maxcos = __fmul_rn(dp[0], il[0]);
        int main_element = 0;
        for (int i = 1; i < 4; ++i) {
            float cosalpha = __fmul_rn(dp[i], il[i]);
            if (cosalpha > maxcos) { main_element = i; maxcos = cosalpha; }
        }
        {
            float lead_x = __fmul_rn(vx[main_element], il[main_element]);
            float lead_y = __fmul_rn(vy[main_element], il[main_element]);
            switch (main_element) {
            case 0: base_a = lead_x; base_b = lead_y; break;
            case 1: base_a = lead_y; base_b = -lead_x; break;
            case 2: base_a = -lead_x; base_b = -lead_y; break;
            default: base_a = -lead_y; base_b = lead_x; break;
            }
        }
        // unrolled select keeps seq/vx/vy/il in registers (no dynamically indexed arrays)
        for (int i = 0; i < 4; i++)
            if (i == main_element) {
                int sidx = seq[i] + 1;
                if (sidx == n) sidx = 0;
                seq[i] = sidx;
                hv_vect(hv, sidx, &vx[i], &vy[i], &il[i]);
            }
        {
            float dx = __fsub_rn(hv.px(seq[1]), hv.px(seq[3]));
            float dy = __fsub_rn(hv.py(seq[1]), hv.py(seq[3]));
            float width = __fadd_rn(__fmul_rn(dx, base_a), __fmul_rn(dy, base_b));
            dx = __fsub_rn(hv.px(seq[2]), hv.px(seq[0]));
            dy = __fsub_rn(hv.py(seq[2]), hv.py(seq[0]));
            float height = __fadd_rn(__fmul_rn(-dx, base_b), __fmul_rn(dy, base_a));
            float area = __fmul_rn(width, height);
            if (area <= minarea) {
                minarea = area;
                buf_i0 = seq[3]; buf1 = base_a; buf2 = width; buf3 = base_b; buf4 = height;
                buf_i5 = seq[0];
            }
        }
    }
    {
        float A1 = buf1, B1 = buf3, A2 = -buf3, B2 = buf1;
        float C1 = __fadd_rn(__fmul_rn(A1, hv.px(buf_i0)), __fmul_rn(hv.py(buf_i0), B1));
        float C2 = __fadd_rn(__fmul_rn(A2, hv.px(buf_i5)), __fmul_rn(hv.py(buf_i5), B2));
        float idet = __fdiv_rn(1.f, __fsub_rn(__fmul_rn(A1, B2), __fmul_rn(A2, B1)));
        float px = __fmul_rn(__fsub_rn(__fmul_rn(C1, B2), __fmul_rn(C2, B1)), idet);
        float py = __fmul_rn(__fsub_rn(__fmul_rn(A1, C2), __fmul_rn(A2, C1)), idet);
        out[0] = px; out[1] = py;
        out[2] = __fmul_rn(A1, buf2); out[3] = __fmul_rn(B1, buf2);
        out[4] = __fmul_rn(A2, buf4); out[5] = __fmul_rn(B2, buf4);
    }
}

template <class HV>
__device__ void rotating_calipers_dev(const HV &hv, float *out) {
    int n = hv.n;
    int left = 0, bottom = 0, right = 0, top = 0;
    float left_x, right_x, top_y, bottom_y;
    left_x = right_x = hv.px(0);
    top_y = bottom_y = hv.py(0);
    for (int i = 0; i < n; i++) {
        float x = hv.px(i), y = hv.py(i);
        if (x < left_x) { left_x = x; left = i; }
        if (x > right_x) { right_x = x; right = i; }
        if (y > top_y) { top_y = y; top = i; }
        if (y < bottom_y) { bottom_y = y; bottom = i; }
    }
    rotating_calipers_core(hv, left, bottom, right, top, out);
}

// What the wave-per-key kernel works out with all its lanes before the (sequential) calipers: the scans over the hull.
struct HullPre {
    int left, bottom, right, top;   // first vertex with the smallest x / smallest y / largest x / largest y
    long long a2;                   // twice the signed area (shoelace)
    int xmin, xmax, ymin, ymax;
};

__device__ __forceinline__ long long cross_i(int2 o, int2 a, int2 b) {
    return (long long)(a.x - o.x) * (b.y - o.y) - (long long)(a.y - o.y) * (b.x - o.x);
}

// minAreaRect of a hull -> lfd's filter -> boxPoints -> truncated quad appended to the slot's list
template <class HV>
__device__ __forceinline__ void rect_from_hull(const HV &hv, double minLen, double lwTresh, int *cnt, int *quads,
                                               size_t quad_base, bool writer, const HullPre *pre = nullptr) {
    float cx = 0, cy = 0, sw = 0, sh = 0, angle = 0;
    if (hv.n > 2) {
        // Cheap certain rejection before the calipers: the rectangle contains the hull (area sw * sh >= A) and
        // its longer side is a projection extent of the hull (<= its diameter <= the bounding-box diagonal D),
        // so length / width = length^2 / (sw * sh) <= D^2 / A.  A roundish contour (the Canny ring of a star,
        // D^2 / A ~ 1.3) can never pass `length / width > lwTresh`; the 0.1 % margin is three orders of
        // magnitude above the float32 noise of the exact computation.  Integer shoelace, exact.
        {
            int xmin, xmax, ymin, ymax;
            long long a2 = 0;
            if (pre) { xmin = pre->xmin; xmax = pre->xmax; ymin = pre->ymin; ymax = pre->ymax; a2 = pre->a2; }
            else {
                int2 p0 = hv.at(0);
                xmin = p0.x; xmax = p0.x; ymin = p0.y; ymax = p0.y;
                int2 prev = p0;
                for (int i = 1; i < hv.n; i++) {
                    int2 q = hv.at(i);
                    a2 += (long long)prev.x * q.y - (long long)q.x * prev.y;
                    xmin = min(xmin, q.x); xmax = max(xmax, q.x); ymin = min(ymin, q.y); ymax = max(ymax, q.y);
                    prev = q;
                }
                a2 += (long long)prev.x * p0.y - (long long)p0.x * prev.y;
            }
            if (a2 < 0) a2 = -a2;
            double d2 = (double)(xmax - xmin) * (xmax - xmin) + (double)(ymax - ymin) * (ymax - ymin);
            if (a2 > 0 && lwTresh > 0 && 2.0 * d2 < lwTresh * (double)a2 * 0.999) return;
        }
        float out[6];
        if (pre) rotating_calipers_core(hv, pre->left, pre->bottom, pre->right, pre->top, out);
        else rotating_calipers_dev(hv, out);
        cx = __fadd_rn(out[0], __fmul_rn(__fadd_rn(out[2], out[4]), 0.5f));
        cy = __fadd_rn(out[1], __fmul_rn(__fadd_rn(out[3], out[5]), 0.5f));
        sw = (float)sqrt((double)out[2] * out[2] + (double)out[3] * out[3]);
        sh = (float)sqrt((double)out[4] * out[4] + (double)out[5] * out[5]);
        angle = (float)atan2((double)out[3], (double)out[2]);
    } else if (hv.n == 2) {
        float x0 = hv.px(0), y0 = hv.py(0), x1 = hv.px(1), y1 = hv.py(1);
        cx = __fmul_rn(__fadd_rn(x0, x1), 0.5f);
        cy = __fmul_rn(__fadd_rn(y0, y1), 0.5f);
        double dx = (double)__fsub_rn(x1, x0), dy = (double)__fsub_rn(y1, y0);
        sw = (float)sqrt(dx * dx + dy * dy);
        sh = 0;
        angle = (float)atan2(dy, dx);
    } else {
        cx = hv.px(0); cy = hv.py(0);
    }
    angle = (float)((double)__fmul_rn(angle, 180.0f) / 3.1415926535897932384626433832795);
    double length, width;
    if (sw > sh) { length = sw; width = sh; } else { width = sw; length = sh; }
    if (!(length > minLen && width > minLen)) return;
    if (!(length / width > lwTresh)) return;
    // RotatedRect::points
    double ang = (double)angle * 3.1415926535897932384626433832795 / 180.;
    float b = __fmul_rn((float)cos(ang), 0.5f);
    float a_ = __fmul_rn((float)sin(ang), 0.5f);
    float bx[8];
    bx[0] = __fsub_rn(__fsub_rn(cx, __fmul_rn(a_, sh)), __fmul_rn(b, sw));
    bx[1] = __fsub_rn(__fadd_rn(cy, __fmul_rn(b, sh)), __fmul_rn(a_, sw));
    bx[2] = __fsub_rn(__fadd_rn(cx, __fmul_rn(a_, sh)), __fmul_rn(b, sw));
    bx[3] = __fsub_rn(__fsub_rn(cy, __fmul_rn(b, sh)), __fmul_rn(a_, sw));
    bx[4] = __fsub_rn(__fmul_rn(2.f, cx), bx[0]);
    bx[5] = __fsub_rn(__fmul_rn(2.f, cy), bx[1]);
    bx[6] = __fsub_rn(__fmul_rn(2.f, cx), bx[2]);
    bx[7] = __fsub_rn(__fmul_rn(2.f, cy), bx[3]);
    if (!writer) return;
    int qi = atomicAdd(&cnt[C_NQUADS], 1);
    cnt[C_DETECT] = 1;
    int *qd = quads + (quad_base + qi) * 8;
    for (int k = 0; k < 8; k++) qd[k] = (int)bx[k]; // np.int32: truncation toward zero
}

// One thread per key (grid-stride): hull -> minAreaRect -> filter -> boxPoints -> quad.
// rects (optional, 5 floats per key in key order) is for the per-operator parity test.
__global__ void __launch_bounds__(64)
k_rects(const int4 *keys, const int2 *rowext, int2 *hullbuf, int *quads, int *counters, int h,
        int w, int key_cap, int slot_cap, double minLen, double lwTresh, const int *active) {
    int g = blockIdx.y;
    if (slot_off(active, counters, g)) return;
    int *cnt = counters + g * C_COUNT;
    int nkeys = min(cnt[C_NKEYS], key_cap);
    const int4 *kg = keys + (size_t)g * key_cap;
    const int2 *re = rowext + (size_t)g * slot_cap;
    // per thread in LDS: the key's row extremes (fetched with independent loads first) and the
    // two chain stacks -- the hull walk is a chain of dependent accesses, which in global memory
    // made this kernel pure latency
    // (lane-interleaved: element k of lane t sits at k * 64 + t.  With a block of 3 x 16 int2 per lane -- 96 dwords, a multiple of
    // the 32 banks -- every lane's element k fell into the same bank: the counters showed five times as many LDS clocks lost to
    // bank conflicts as spent on the accesses themselves: profiles/r04_util_sdss.json, k_rects)
    __shared__ int2 stk[64 * 3 * SMALL_KEY_ROWS];
    int2 *ext = stk + threadIdx.x, *c1 = ext + 64 * SMALL_KEY_ROWS, *c2 = c1 + 64 * SMALL_KEY_ROWS;
#define LANE_AT(p_, k_) (p_)[(k_) * 64]
    (void)hullbuf;
    for (int ki = blockIdx.x * blockDim.x + threadIdx.x; ki < nkeys; ki += gridDim.x * blockDim.x) {
        int4 key = kg[ki];
        int extent = key.y & ~KEY_HOLE_BIT, ymin = key.z, base = key.w;
        if (extent > SMALL_KEY_ROWS) continue; // k_rects_big
        for (int r = 0; r < extent; r++) LANE_AT(ext, r) = re[base + r];
        int n1 = 0, n2 = 0;
        for (int r = 0; r < extent; r++) {
            int2 e = LANE_AT(ext, r);
            if (e.x > e.y) continue;
            int2 p = make_int2(e.x, ymin + r);
            while (n1 >= 2 && cross_i(LANE_AT(c1, n1 - 2), LANE_AT(c1, n1 - 1), p) >= 0) n1--;
            LANE_AT(c1, n1++) = p;
        }
        for (int r = extent - 1; r >= 0; r--) {
            int2 e = LANE_AT(ext, r);
            if (e.x > e.y) continue;
            int2 p = make_int2(e.y, ymin + r);
            while (n2 >= 2 && cross_i(LANE_AT(c2, n2 - 2), LANE_AT(c2, n2 - 1), p) >= 0) n2--;
            LANE_AT(c2, n2++) = p;
        }
        if (n1 == 0) continue;
        // drop the vertices the two chains share at the bottom and at the top
        int a = 0, bnd = n2;
        if (LANE_AT(c2, 0).x == LANE_AT(c1, n1 - 1).x && LANE_AT(c2, 0).y == LANE_AT(c1, n1 - 1).y) a = 1;
        if (bnd > a && LANE_AT(c2, bnd - 1).x == LANE_AT(c1, 0).x && LANE_AT(c2, bnd - 1).y == LANE_AT(c1, 0).y) bnd--;
        LaneHullView hv;
        hv.c1 = c1; hv.c2 = c2 + a * 64; hv.n1 = n1; hv.n = n1 + (bnd - a); hv.s0 = 0;
        if (hv.n < 1) continue;
        int s0 = 0;
        int2 best = hv.raw(0);
        for (int i = 1; i < hv.n; i++) {
            int2 v = hv.raw(i);
            if (v.x < best.x || (v.x == best.x && v.y < best.y)) { best = v; s0 = i; }
        }
        hv.s0 = s0;
        rect_from_hull(hv, minLen, lwTresh, cnt, quads, (size_t)g * key_cap, true);
    }
#undef LANE_AT
}


// ---- big keys: one wave per key, hull candidates filtered in parallel in LDS -----------------
// A chain candidate p with neighbours a, b (in chain order) that does not make a strict right
// turn (cross >= 0) lies on or inside the hull and can never be a strict hull vertex; dropping
// all such points at once and repeating until nothing changes leaves exactly the strictly
// convex chain the sequential monotone-chain scan produces (the survivors contain every hull
// vertex and form a strictly convex polyline, which cannot hold a non-vertex).
struct LdsHullView {
    const int2 *c1, *c2;
    int n1, n, s0;
    __device__ __forceinline__ int2 raw(int i) const { return i < n1 ? c1[i] : c2[i - n1]; }
    __device__ __forceinline__ int2 at(int i) const {
        int k = i + s0;
        if (k >= n) k -= n;
        return raw(k);
    }
    __device__ __forceinline__ float px(int i) const { return (float)at(i).x; }
    __device__ __forceinline__ float py(int i) const { return (float)at(i).y; }
};

// compacts the chain held in A (n points) into strictly convex form; returns the buffer holding it
__device__ __forceinline__ int2 *filter_chain(int2 *A, int2 *B, int *n_io) {
    int n = *n_io, lane = lfd_lane();
    for (int pass = 0; pass < 4096 && n > 2; pass++) {
        int m = 0;
        for (int i0 = 0; i0 < n; i0 += 64) {
            int i = i0 + lane;
            bool keep = false;
            int2 p = make_int2(0, 0);
            if (i < n) {
                p = A[i];
                keep = (i == 0) || (i == n - 1) || cross_i(A[i - 1], p, A[i + 1]) < 0;
            }
            u64 bal = __ballot(keep);
            if (keep) B[m + __popcll(bal & ((1ull << lane) - 1ull))] = p;
            m += __popcll(bal);
        }
        __syncthreads(); // one wave per block: orders the LDS writes before the next pass reads them
        int2 *t = A; A = B; B = t;
        if (m == n) break;
        n = m;
    }
    *n_io = n;
    return A;
}

// The hull of a wave-per-key key laid out for the sequential calipers: vertices rotated so that vertex 0 is (min x, then min y),
// and every edge's vector and reciprocal length (hv_vect's double-precision square root and division, the expensive part of a
// calipers step) evaluated once, by all lanes side by side, with the expressions of hv_vect -- same bits, no dependent chain.
struct PreHullView {
    const int2 *P;     // n vertices
    const float *E;    // n x (vx, vy, 1 / length) of the edge i -> i + 1
    int n;
    __device__ __forceinline__ int2 at(int i) const { return P[i]; }
    __device__ __forceinline__ float px(int i) const { return (float)P[i].x; }
    __device__ __forceinline__ float py(int i) const { return (float)P[i].y; }
};
__device__ __forceinline__ void hv_vect(const PreHullView &hv, int i, float *vx, float *vy, float *inv) {
    *vx = hv.E[3 * i]; *vy = hv.E[3 * i + 1]; *inv = hv.E[3 * i + 2];
}

// lane-parallel argmin / argmax over (value, index) with the lowest index on equal values
__device__ __forceinline__ void wave_arg(int &v, int &i, bool want_max) {
    for (int o = 32; o > 0; o >>= 1) {
        int ov = __shfl_xor(v, o), oi = __shfl_xor(i, o);
        bool take = want_max ? (ov > v || (ov == v && oi < i)) : (ov < v || (ov == v && oi < i));
        if (take) { v = ov; i = oi; }
    }
}

// Fills P (LDS, at least n int2) from the raw hull and returns the scans of rect_from_hull / the calipers.
template <class HV>
__device__ __forceinline__ HullPre wave_prepare_hull(const HV &raw, int2 *P) {
    const int n = raw.n, lane = lfd_lane();
    // vertex 0: smallest x, then smallest y (vertices are distinct)
    int bx = 0x7fffffff, by = 0x7fffffff, bi = 0x7fffffff;
    for (int i = lane; i < n; i += 64) {
        int2 v = raw.raw(i);
        if (v.x < bx || (v.x == bx && v.y < by)) { bx = v.x; by = v.y; bi = i; }
    }
    for (int o = 32; o > 0; o >>= 1) {
        int ox = __shfl_xor(bx, o), oy = __shfl_xor(by, o), oi = __shfl_xor(bi, o);
        if (ox < bx || (ox == bx && oy < by)) { bx = ox; by = oy; bi = oi; }
    }
    const int s0 = bi;
    for (int i = lane; i < n; i += 64) {
        int k = i + s0;
        if (k >= n) k -= n;
        P[i] = raw.raw(k);
    }
    __syncthreads(); // (one wave per block)
    HullPre pre;
    int lv = 0x7fffffff, li = 0x7fffffff, rv = -0x7fffffff - 1, ri = 0x7fffffff, tv = -0x7fffffff - 1, ti = 0x7fffffff, bv = 0x7fffffff, bti = 0x7fffffff;
    long long a2 = 0;
    for (int i = lane; i < n; i += 64) {
        int j = (i + 1 < n) ? i + 1 : 0;
        int2 a = P[i], b = P[j];
        a2 += (long long)a.x * b.y - (long long)b.x * a.y;
        if (a.x < lv) { lv = a.x; li = i; }   // (i increases per lane: strict compares keep the lane's first index)
        if (a.x > rv) { rv = a.x; ri = i; }
        if (a.y > tv) { tv = a.y; ti = i; }
        if (a.y < bv) { bv = a.y; bti = i; }
    }
    wave_arg(lv, li, false);
    wave_arg(rv, ri, true);
    wave_arg(tv, ti, true);
    wave_arg(bv, bti, false);
    for (int o = 32; o > 0; o >>= 1) a2 += __shfl_xor(a2, o);
    pre.left = li; pre.right = ri; pre.top = ti; pre.bottom = bti;
    pre.a2 = a2;
    pre.xmin = lv; pre.xmax = rv; pre.ymax = tv; pre.ymin = bv;
    __syncthreads();
    return pre;
}

// E (LDS, 3 n floats): every edge's vector and reciprocal length, hv_vect's expressions, one edge per lane and round
__device__ __forceinline__ void wave_edges(const int2 *P, float *E, int n) {
    for (int i = lfd_lane(); i < n; i += 64) {
        int j = (i + 1 < n) ? i + 1 : 0;
        int2 a = P[i], b = P[j];
        double dx = (double)__fsub_rn((float)b.x, (float)a.x);
        double dy = (double)__fsub_rn((float)b.y, (float)a.y);
        E[3 * i] = (float)dx;
        E[3 * i + 1] = (float)dy;
        E[3 * i + 2] = (float)(1. / sqrt(dx * dx + dy * dy));
    }
    __syncthreads();
}

__global__ void __launch_bounds__(64)
k_rects_big(const int4 *keys, const int *bigkeys, int cidx, const int2 *rowext, int *quads, int *counters, int h, int w,
            int key_cap, int slot_cap, int cap, double minLen, double lwTresh, const int *active, int wave_prep, int ext_lo, int skip_above) {
    // (ext_lo, skip_above: the tall keys go in two launches over the same list -- up to `cap` rows with a small LDS footprint and
    // many workgroups per CU, the few taller ones with the full-height footprint -- each skipping the other's keys)
    int g = blockIdx.y;
    if (slot_off(active, counters, g)) return;
    extern __shared__ int2 lds_pts[]; // 4 x cap
    int *cnt = counters + g * C_COUNT;
    int nbig = cnt[cidx];
    const int4 *kg = keys + (size_t)g * key_cap;
    const int2 *re = rowext + (size_t)g * slot_cap;
    int lane = lfd_lane();
    for (int bi = blockIdx.x; bi < nbig; bi += gridDim.x) {
        int4 key = kg[bigkeys[(size_t)g * key_cap + bi]];
        int extent = key.y & ~KEY_HOLE_BIT, ymin = key.z, base = key.w;
        if (extent <= ext_lo) continue;
        if (extent > cap) { if (!skip_above && lane == 0) cnt[C_OVERFLOW] = 1; continue; }
        int2 *P0 = lds_pts, *P1 = lds_pts + cap, *P2 = lds_pts + 2 * cap, *P3 = lds_pts + 3 * cap;
        __syncthreads();
        // the key's rows into P3 first, eight loads per lane in flight (a tall key is 24 rounds of 64 rows; one dependent
        // global load per round and chain was most of this kernel's time)
        for (int r0 = 0; r0 < extent; r0 += 512) {
            int2 v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                int r = r0 + u * 64 + lane;
                v[u] = r < extent ? re[base + r] : make_int2(1, 0);
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                int r = r0 + u * 64 + lane;
                if (r < extent) P3[r] = v[u];
            }
        }
        __syncthreads();
        // chain 1: left-most pixel of every row, rows increasing
        int n1 = 0;
        for (int r0 = 0; r0 < extent; r0 += 64) {
            int r = r0 + lane;
            int2 e = make_int2(1, 0);
            if (r < extent) e = P3[r];
            bool ok = e.x <= e.y;
            u64 bal = __ballot(ok);
            if (ok) P0[n1 + __popcll(bal & ((1ull << lane) - 1ull))] = make_int2(e.x, ymin + r);
            n1 += __popcll(bal);
        }
        // chain 2: right-most pixel of every row, rows decreasing
        int n2 = 0;
        for (int r0 = 0; r0 < extent; r0 += 64) {
            int r = extent - 1 - (r0 + lane);
            int2 e = make_int2(1, 0);
            if (r >= 0) e = P3[r];
            bool ok = e.x <= e.y;
            u64 bal = __ballot(ok);
            if (ok) P2[n2 + __popcll(bal & ((1ull << lane) - 1ull))] = make_int2(e.y, ymin + r);
            n2 += __popcll(bal);
        }
        __syncthreads();
        if (n1 == 0) continue;
        int2 *c1 = filter_chain(P0, P1, &n1);
        int2 *c2 = filter_chain(P2, P3, &n2);
        int a = 0, bnd = n2;
        if (c2[0].x == c1[n1 - 1].x && c2[0].y == c1[n1 - 1].y) a = 1;
        if (bnd > a && c2[bnd - 1].x == c1[0].x && c2[bnd - 1].y == c1[0].y) bnd--;
        LdsHullView hv;
        hv.c1 = c1; hv.c2 = c2 + a; hv.n1 = n1; hv.n = n1 + (bnd - a); hv.s0 = 0;
        // the two buffers the chains did not end up in hold the prepared hull when it fits (it does unless a key is nearly convex
        // over its whole height: 3 floats per edge in 2 * cap floats)
        int2 *F1 = (c1 == P0) ? P1 : P0, *F2 = (c2 == P2) ? P3 : P2;
        if (hv.n > 2 && 3 * hv.n <= 2 * cap && wave_prep) {
            HullPre pre = wave_prepare_hull(hv, F1);
            { // rect_from_hull's certain rejection (roundish contours: most keys), before the edges are worked out
                long long A2 = pre.a2 < 0 ? -pre.a2 : pre.a2;
                double d2 = (double)(pre.xmax - pre.xmin) * (pre.xmax - pre.xmin) + (double)(pre.ymax - pre.ymin) * (pre.ymax - pre.ymin);
                if (A2 > 0 && lwTresh > 0 && 2.0 * d2 < lwTresh * (double)A2 * 0.999) continue;
            }
            wave_edges(F1, (float *)F2, hv.n);
            PreHullView pv;
            pv.P = F1; pv.E = (const float *)F2; pv.n = hv.n;
            rect_from_hull(pv, minLen, lwTresh, cnt, quads, (size_t)g * key_cap, lane == 0, &pre);
            continue;
        }
        int s0 = 0;
        int2 best = hv.raw(0);
        for (int i = 1; i < hv.n; i++) {
            int2 v = hv.raw(i);
            if (v.x < best.x || (v.x == best.x && v.y < best.y)) { best = v; s0 = i; }
        }
        hv.s0 = s0;
        // every lane evaluates the same rectangle (uniform control flow, LDS broadcasts); lane 0 writes
        rect_from_hull(hv, minLen, lwTresh, cnt, quads, (size_t)g * key_cap, lane == 0);
    }
}

// ---- cv2.fillPoly of one quad into bit rows ----------------------------------------------
__device__ __forceinline__ void set_span(u64 *rowbits, int xa, int xb) {
    for (int k = xa >> 6; k <= (xb >> 6); k++) {
        int lo = (k == (xa >> 6)) ? (xa & 63) : 0, hi = (k == (xb >> 6)) ? (xb & 63) : 63;
        u64 msk = (~0ull << lo) & (~0ull >> (63 - hi));
        atomicOr(&rowbits[k], msk);
    }
}

__device__ bool clip_line_dev(long long width, long long height, long long *x1, long long *y1,
                              long long *x2, long long *y2) {
    int c1, c2;
    long long right = width - 1, bottom = height - 1;
    if (width <= 0 || height <= 0) return false;
    c1 = (*x1 < 0) + (*x1 > right) * 2 + (*y1 < 0) * 4 + (*y1 > bottom) * 8;
    c2 = (*x2 < 0) + (*x2 > right) * 2 + (*y2 < 0) * 4 + (*y2 > bottom) * 8;
    if ((c1 & c2) == 0 && (c1 | c2) != 0) {
        long long a;
        if (c1 & 12) {
            a = c1 < 8 ? 0 : bottom;
            *x1 += (long long)((double)(a - *y1) * (double)(*x2 - *x1) / (double)(*y2 - *y1));
            *y1 = a;
            c1 = (*x1 < 0) + (*x1 > right) * 2;
        }
        if (c2 & 12) {
            a = c2 < 8 ? 0 : bottom;
            *x2 += (long long)((double)(a - *y2) * (double)(*x2 - *x1) / (double)(*y2 - *y1));
            *y2 = a;
            c2 = (*x2 < 0) + (*x2 > right) * 2;
        }
        if ((c1 & c2) == 0 && (c1 | c2) != 0) {
            if (c1) {
                a = c1 == 1 ? 0 : right;
                *y1 += (long long)((double)(a - *x1) * (double)(*y2 - *y1) / (double)(*x2 - *x1));
                *x1 = a;
                c1 = 0;
            }
            if (c2) {
                a = c2 == 1 ? 0 : right;
                *y2 += (long long)((double)(a - *x2) * (double)(*y2 - *y1) / (double)(*x2 - *x1));
                *x2 = a;
                c2 = 0;
            }
        }
    }
    return (c1 | c2) == 0;
}

#define FILL_BLOCKS 64

__global__ void __launch_bounds__(256)
k_fill_quads(const int *quads, const int *counters, u64 *box, int h, int w, int key_cap,
             const int *active) {
    int g = blockIdx.y;
    if (slot_off(active, counters, g)) return;
    const int *cnt = counters + g * C_COUNT;
    int nq = min(cnt[C_NQUADS], key_cap);
    int wq = LFD_WQ(w);
    u64 *bg = box + (size_t)g * h * wq;
    for (int qi = blockIdx.x; qi < nq; qi += gridDim.x) {
        const int *v = quads + ((size_t)g * key_cap + qi) * 8;
        long long vx[4], vy[4];
        for (int k = 0; k < 4; k++) { vx[k] = v[2 * k]; vy[k] = v[2 * k + 1]; }
        // outline: Line() = 8-connected Bresenham after clipLine, closed form per step
        for (int e = 0; e < 4; e++) {
            int e0 = (e + 3) & 3;
            long long x1 = vx[e0], y1 = vy[e0], x2 = vx[e], y2 = vy[e];
            bool ok = true;
            if ((unsigned long long)x1 >= (unsigned long long)w || (unsigned long long)x2 >= (unsigned long long)w ||
                (unsigned long long)y1 >= (unsigned long long)h || (unsigned long long)y2 >= (unsigned long long)h)
                ok = clip_line_dev(w, h, &x1, &y1, &x2, &y2);
            if (!ok) continue;
            int dx = (int)(x2 - x1), dy = (int)(y2 - y1);
            int px = (int)x1, py = (int)y1;
            if (dx < 0) { dx = -dx; dy = -dy; px = (int)x2; py = (int)y2; }
            int sy = dy < 0 ? -1 : 1;
            if (dy < 0) dy = -dy;
            bool steep = dy > dx;
            if (steep) { int t = dx; dx = dy; dy = t; }
            int count = dx + 1;
            for (int i = threadIdx.x; i < count; i += 256) {
                // number of minor steps before point i: ceil((2*dy*i - dx) / (2*dx)), >= 0
                int minor = dx ? (int)((2ll * dy * i + dx - 1) / (2ll * dx)) : 0;
                int x = steep ? px + minor : px + i;
                int y = steep ? py + sy * i : py + sy * minor;
                atomicOr(&bg[(size_t)y * wq + (x >> 6)], 1ull << (x & 63));
            }
        }
        // interior: even-odd scan conversion, 16.16 fixed point (FillEdgeCollection)
        int ey0[4], ey1[4], ne = 0;
        long long ex[4], edx[4];
        int ymin = 0x7fffffff, ymax = -0x7fffffff - 1;
        long long xmin = 0x7fffffffffffffffLL, xmax = -1;
        for (int e = 0; e < 4; e++) {
            int e0 = (e + 3) & 3;
            long long p0x = vx[e0] << 16, p0y = vy[e0], p1x = vx[e] << 16, p1y = vy[e];
            if (p0y == p1y) continue;
            if (p0y < p1y) { ey0[ne] = (int)p0y; ey1[ne] = (int)p1y; ex[ne] = p0x; }
            else { ey0[ne] = (int)p1y; ey1[ne] = (int)p0y; ex[ne] = p1x; }
            edx[ne] = (p1x - p0x) / (p1y - p0y);
            long long xe = ex[ne] + (long long)(ey1[ne] - ey0[ne]) * edx[ne];
            ymin = min(ymin, ey0[ne]); ymax = max(ymax, ey1[ne]);
            xmin = min(xmin, min(ex[ne], xe)); xmax = max(xmax, max(ex[ne], xe));
            ne++;
        }
        if (ne < 2) continue;
        if (ymax < 0 || ymin >= h || xmax < 0 || xmin >= ((long long)w << 16)) continue;
        if (ymax > h) ymax = h;
        int ystart = ymin < 0 ? 0 : ymin;
        for (int y = ystart + threadIdx.x; y < ymax; y += 256) {
            long long xs[4];
            int na = 0;
            for (int e = 0; e < 4; e++)
                if (e < ne && ey0[e] <= y && y < ey1[e]) {
                    long long xv = ex[e] + (long long)(y - ey0[e]) * edx[e];
                    int k = na++;
                    while (k > 0 && xs[k - 1] > xv) { xs[k] = xs[k - 1]; k--; }
                    xs[k] = xv;
                }
            for (int k = 0; k + 1 < na; k += 2) {
                int xa = (int)((xs[k] + 65535) >> 16);
                int xb = (int)(xs[k + 1] >> 16);
                if (xa < w && xb >= 0) {
                    if (xa < 0) xa = 0;
                    if (xb >= w) xb = w - 1;
                    if (xa <= xb) set_span(bg + (size_t)y * wq, xa, xb);
                }
            }
        }
    }
}

// Developer check of the three libm calls on the accept / reject path above (minAreaRect's angle, RotatedRect::points):
// the device evaluates exactly the expressions of rect_from_hull on caller-supplied operands, so that a test can compare
// the float32 results with the host libm's on millions of inputs (tests/test_gpu_stages.py).
__global__ void __launch_bounds__(256)
k_debug_trig(const double *y, const double *x, float *angle_deg, float *cos_half, float *sin_half, int n) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float angle = (float)atan2(y[i], x[i]);
    angle = (float)((double)__fmul_rn(angle, 180.0f) / 3.1415926535897932384626433832795);
    double ang = (double)angle * 3.1415926535897932384626433832795 / 180.;
    angle_deg[i] = angle;
    cos_half[i] = __fmul_rn((float)cos(ang), 0.5f);
    sin_half[i] = __fmul_rn((float)sin(ang), 0.5f);
}

