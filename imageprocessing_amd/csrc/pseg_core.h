// pseg_core.h -- per-thread bodies of the sparse (run-based) segmentation kernels.
//
// Everything after the dense streaming kernels works on the 1-bit threshold plane: foreground
// runs per row, union-find over runs (8-connected foreground, 4-connected background incl. the
// zero padding the reference's findContours assumes), the enclosure tree (external component ->
// hole -> nested component ...), and per-object integer accumulators.  The bodies are plain
// functions of a thread index so that the HIP kernels in piscoseg.hip are one-line wrappers (the
// host versions of the atomics below let the same bodies be compiled and stepped on a CPU).
//
// Reference semantics restated:
//   cv.findContours(RETR_EXTERNAL)      detection.py:204  -> external components = components
//                                        adjacent to the frame-connected background
//   cv.contourArea                      detection.py:208  -> Q4 + Q3/2 over 2x2 quads of the
//                                        hole-filled object
//   cv.boundingRect                     detection.py:212  -> extents of the component's runs
//   cv.drawContours(FILLED)             detection.py:245  -> component + everything it encloses
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define PSEG_HD __host__ __device__ inline
#else
#define PSEG_HD inline
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define PSEG_ATOMIC_MIN(p, v) atomicMin((p), (v))
#define PSEG_ATOMIC_MAX(p, v) atomicMax((p), (v))
#define PSEG_ATOMIC_ADD(p, v) atomicAdd((p), (v))
#define PSEG_POPC64(v) __popcll(v)
#define PSEG_CTZ64(v) (__ffsll((unsigned long long)(v)) - 1)
#else
template <class T> static inline T pseg_amin(T *p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <class T> static inline T pseg_amax(T *p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T> static inline T pseg_aadd(T *p, T v) { T o = *p; *p = o + v; return o; }
#define PSEG_ATOMIC_MIN(p, v) pseg_amin((p), (v))
#define PSEG_ATOMIC_MAX(p, v) pseg_amax((p), (v))
#define PSEG_ATOMIC_ADD(p, v) pseg_aadd((p), (v))
#define PSEG_POPC64(v) __builtin_popcountll(v)
#define PSEG_CTZ64(v) __builtin_ctzll(v)
#endif

// Device-resident state of one detection call (all pointers device memory).
struct SegState {
    int F, H, W, WW;                 // frames, rows, cols, 64-bit words per row
    const uint64_t *bits;            // [F][H][WW] threshold plane, bit b of word w = pixel 64w+b
    const uint8_t *valid;            // [F] 0 = corrupt frame (treated as empty)
    uint32_t *rowcnt;                // [F*H] foreground runs per row
    uint32_t *rowoff;                // [F*(H+1)] global index of the first run of every row (+ one end entry per frame)
    uint32_t *framebase;             // [F+1] exclusive scan of runs per frame
    // per foreground run (global index r = framebase[f] + rowoff[f][y] + j)
    uint32_t *run_x;                 // xs | xe << 16
    uint32_t *run_fy;                // y | f << 16
    int32_t *fg_parent;              // union-find, root = smallest (raster-first) run
    int32_t *ext;                    // root run of the enclosing EXTERNAL component
    int32_t *gapl;                   // external object whose filled region holds the gap left of the run, -1 = none
    // per background element (b = r + f*H + y for the gap left of run r; one extra per row)
    int32_t *bg_parent;
    int32_t *bg_left;                // foreground run immediately left of the gap, -1 for j = 0
    // per object, indexed by its root run
    int32_t *minx, *maxx, *maxy;        // (min y = row of the root run)
    int32_t *area2;                  // 2*Q4 + Q3 = 2 * contourArea
    int32_t *extrank, *keptrank, *savedrank;   // exclusive ranks inside the frame
};

PSEG_HD int seg_rowstart(const SegState &S, int f, int y) { return (int)S.rowoff[f * (S.H + 1) + y]; }
PSEG_HD int seg_rowcount(const SegState &S, int f, int y) { return (int)(S.rowoff[f * (S.H + 1) + y + 1] - S.rowoff[f * (S.H + 1) + y]); }
PSEG_HD int seg_bgbase(const SegState &S, int f, int y) { return seg_rowstart(S, f, y) + f * S.H + y; }
PSEG_HD int seg_b0(const SegState &S, int f) { return (int)S.framebase[f] + f * S.H; }   // smallest background index of frame f
PSEG_HD int run_xs(const SegState &S, int r) { return (int)(S.run_x[r] & 0xffffu); }
PSEG_HD int run_xe(const SegState &S, int r) { return (int)(S.run_x[r] >> 16); }

// Runs are extracted in one pass over the plane (k_runs_band in piscoseg.hip: starts = v & ~((v << 1) | carry) per 64-bit
// word, a look-back scan over the bands gives every row its first run index), or in two for very wide rows (k_runs).

// ---- union-find (link the larger root under the smaller one) -------------------------------
PSEG_HD int uf_find(const int32_t *parent, int a)
{
    int p = parent[a];
    while (p != a) { a = p; p = parent[a]; }
    return a;
}
// find with path halving: every visited node is re-pointed to its grandparent.  The plain stores race
// with other threads' atomicMin on roots only in a benign way: a node's parent is always replaced by
// one of its current ancestors (values only decrease along a chain), never by a foreign index.
PSEG_HD int uf_find_halve(int32_t *parent, int a)
{
    int p = parent[a];
    while (p != a) {
        int g = parent[p];
        if (g != p) parent[a] = g;
        a = p; p = g;
    }
    return a;
}
PSEG_HD void uf_union(int32_t *parent, int a, int b)
{
    for (int guard = 0; guard < (1 << 30); guard++) {
        a = uf_find_halve(parent, a);
        b = uf_find_halve(parent, b);
        if (a == b) return;
        int hi = a > b ? a : b, lo = a > b ? b : a;
        int old = PSEG_ATOMIC_MIN(&parent[hi], lo);
        if (old == hi) return;
        a = old; b = lo;
    }
}

// thread t < nfg: run t and the gap left of it; thread t >= nfg: trailing gap of row t - nfg.
// Gaps that touch the zero padding (first / last gap of a row, every gap of the first / last row) start
// with the frame's smallest background index as parent: the frame-connected background needs no unions.
PSEG_HD void init_body(const SegState &S, int t, int nfg)
{
    if (t < nfg) {
        int y = (int)(S.run_fy[t] & 0xffffu), f = (int)(S.run_fy[t] >> 16);
        S.fg_parent[t] = t;
        S.minx[t] = 0x7fffffff; S.maxx[t] = -1; S.maxy[t] = -1;
        S.area2[t] = 0;
        int j = t - seg_rowstart(S, f, y), b = t + f * S.H + y;
        S.bg_parent[b] = (j == 0 || y == 0 || y == S.H - 1) ? seg_b0(S, f) : b;
        S.bg_left[b] = j == 0 ? -1 : t - 1;
    } else {
        int fy = t - nfg, f = fy / S.H, y = fy % S.H;
        int rs = seg_rowstart(S, f, y), n = seg_rowcount(S, f, y);
        int b = rs + f * S.H + y + n;
        S.bg_parent[b] = seg_b0(S, f);
        S.bg_left[b] = n == 0 ? -1 : rs + n - 1;
    }
}

// ---- 3. foreground unions, 8-connectivity (thread per run) -----------------------------------
// bandh > 0: only rows on a band boundary (y % bandh == 0) -- the rows inside a band were united in LDS
PSEG_HD void fg_union_body(const SegState &S, int r, int bandh = 0)
{
    int y = (int)(S.run_fy[r] & 0xffffu), f = (int)(S.run_fy[r] >> 16);
    if (y == 0 || (bandh > 0 && y % bandh != 0)) return;
    int p0 = seg_rowstart(S, f, y - 1), p1 = seg_rowstart(S, f, y);
    if (p0 == p1) return;
    int lo = run_xs(S, r) - 1, hi = run_xe(S, r) + 1;
    int a = p0, b = p1;                                  // first q with xe_q >= lo
    while (a < b) { int m = (a + b) >> 1; if (run_xe(S, m) >= lo) b = m; else a = m + 1; }
    for (int q = a; q < p1 && run_xs(S, q) <= hi; q++) uf_union(S.fg_parent, r, q);
}

// ---- 4. background unions, 4-connectivity, frame padded with zeros ---------------------------
// Element j of row (f,y) is the gap left of foreground run j (j = n is the gap after the last
// run); gaps 0 and n reach the padding columns x = -1 and x = W.
PSEG_HD void bg_union_elem(const SegState &S, int f, int y, int j)
{
    int rs = seg_rowstart(S, f, y), n = seg_rowcount(S, f, y);
    int bb = rs + f * S.H + y, b = bb + j;
    int a = j == 0 ? -1 : run_xe(S, rs + j - 1) + 1;
    int e = j == n ? S.W : run_xs(S, rs + j) - 1;
    if (y == 0) return;                                  // (padding-adjacent gaps were attached to B0 by init_body)
    int ps = seg_rowstart(S, f, y - 1), pn = rs - ps, pb = ps + f * S.H + (y - 1);
    // first gap j' of the previous row whose right end (xs_{j'} - 1, or W for j' = pn) >= a
    int lo = 0, hi = pn;
    while (lo < hi) { int m = (lo + hi) >> 1; if (run_xs(S, ps + m) >= a + 1) hi = m; else lo = m + 1; }
    for (int q = lo; q <= pn; q++) {
        int qa = q == 0 ? -1 : run_xe(S, ps + q - 1) + 1;
        if (qa > e) break;
        uf_union(S.bg_parent, b, pb + q);
    }
}
// thread t < nfg : gap left of run t ; thread t >= nfg : trailing gap of row (t - nfg)
PSEG_HD void bg_union_body(const SegState &S, int t, int nfg, int bandh = 0)
{
    if (t < nfg) {
        int y = (int)(S.run_fy[t] & 0xffffu), f = (int)(S.run_fy[t] >> 16);
        if (bandh > 0 && y % bandh != 0) return;
        bg_union_elem(S, f, y, t - seg_rowstart(S, f, y));
    } else {
        int fy = t - nfg, f = fy / S.H, y = fy % S.H;
        if (bandh > 0 && y % bandh != 0) return;
        bg_union_elem(S, f, y, seg_rowcount(S, f, y));
    }
}

// ---- 5. enclosure tree: external ancestor of every run (k_flatten_resolve in piscoseg.hip) -----
// From a component's root run R: the background element around R's raster-first pixel is either the
// frame-connected background (R is external) or a hole, whose owner is the component of the run left of the
// hole's root gap; repeat from there.
// object id of a background element: -1 = outside every object
PSEG_HD int bg_gid(const SegState &S, int f, int b)
{
    int hb = S.bg_parent[b];
    if (hb == seg_b0(S, f)) return -1;
    return S.ext[S.bg_left[hb]];
}

// ---- 6. bounding boxes: min x / max x / max y per external object (min y is the root's row).
// k_bbox first combines the runs of a block in an LDS hash table, then calls this once per (block, object).
PSEG_HD void bbox_flush(const SegState &S, int E, int xs, int xe, int y)
{
    // monotone values: a (possibly stale) read that already satisfies the bound makes the atomic redundant
    if (xs < S.minx[E]) PSEG_ATOMIC_MIN(&S.minx[E], xs);
    if (xe > S.maxx[E]) PSEG_ATOMIC_MAX(&S.maxx[E], xe);
    if (y > S.maxy[E]) PSEG_ATOMIC_MAX(&S.maxy[E], y);
}

// ---- 7. contour area from 2x2 quads of the filled object ---------------------------------------
// A span = maximal stretch of a row that lies in the hole-filled region of one object: runs of the object (and of
// components nested in it) joined by the gaps it fills; a span begins and ends with a run (a filled gap always has a run
// of the same object on either side).  Thread per run: the run that begins a span walks to the span's end, finds the
// spans of the same object in the next row that overlap it and adds, per pair [a,b] x [c,d],
//   2 * #(columns x with x, x+1 in both)  +  #(2x2 quads with exactly three pixels set)   to area2[object].
// Returns the contribution of run r (0 for runs that do not begin a span) and its object in *obj.
PSEG_HD int area_contrib(const SegState &S, int r, int *obj)
{
    const int id = S.ext[r];
    *obj = id;
    const int y = (int)(S.run_fy[r] & 0xffffu), f = (int)(S.run_fy[r] >> 16);
    if (y + 1 >= S.H) return 0;
    const int rs = seg_rowstart(S, f, y), re = seg_rowstart(S, f, y + 1), ne = re + seg_rowcount(S, f, y + 1);
    if (re == ne) return 0;                              // the next row has no runs
    if (r > rs && S.gapl[r] == id) return 0;             // not the first run of its span
    const int a = run_xs(S, r);
    int e = r;
    while (e + 1 < re && S.gapl[e + 1] == id) e++;
    const int b = run_xe(S, e);
    int lo = re, hi = ne;                                // first run q of the next row with xe >= a
    while (lo < hi) { int m = (lo + hi) >> 1; if (run_xe(S, m) >= a) hi = m; else lo = m + 1; }
    int q = lo;
    while (q > re && S.gapl[q] == S.ext[q]) q--;         // that run may sit inside a span that began further left
    int add = 0;
    while (q < ne) {
        const int qid = S.ext[q];
        const int c = run_xs(S, q);
        if (c > b) break;
        int qe = q;
        while (qe + 1 < ne && S.gapl[qe + 1] == qid) qe++;
        const int d = run_xe(S, qe);
        if (qid == id && d >= a) {
            const int l = a > c ? a : c, h = b < d ? b : d;
            add += 2 * (h - l) + (a <= c - 1) + (d + 1 <= b) + (c <= a - 1) + (b + 1 <= d);
        }
        q = qe + 1;
    }
    return add;
}
PSEG_HD void area_body(const SegState &S, int r)
{
    int id;
    const int add = area_contrib(S, r, &id);
    if (add) PSEG_ATOMIC_ADD(&S.area2[id], add);
}

// classification of a run as the root of an external component
PSEG_HD bool is_ext_root(const SegState &S, int r) { return S.fg_parent[r] == r && S.ext[r] == r; }
