/*
 * pisco_oracle.c -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C restatement of the OpenCV calls that the reference's detection stage makes
 * (reference: /root/reference/detection.py:195-216,245).  Nothing in the product
 * (imageprocessing_amd/) may import, link or call this file; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg do, and only as the checker /
 * timed CPU baseline.
 *
 * PARITY UNPINNED for the OpenCV semantics: the arithmetic lives in the third-party wheel
 * `opencv-python` (unpinned, /root/reference/requirements.txt:7), which is absent from
 * /root/reference and from this image, and the reference ships no tests or golden vectors.
 * The functions below restate the published algorithms of that library:
 *   - cv::findContours(RETR_EXTERNAL, CHAIN_APPROX_SIMPLE): Suzuki & Abe 1985 border following
 *     as the library's sequential raster scanner applies it (zero-padded frame, marks nbd=2 /
 *     -126 for "right side examined", "last marked pixel on this row > 0" test for nested
 *     borders, contours returned in reverse order of discovery).
 *   - cv::contourArea: 0.5*|sum cross| over the vertex list, accumulated in double.
 *   - cv::boundingRect: inclusive extents of the vertex list.
 *   - cv::drawContours(thickness=FILLED): 8-connected outline between consecutive vertices
 *     plus even-odd scanline fill of the vertex polygon with half-open [y0,y1) edges.
 *   - cv::circle(thickness=-1, LINE_8, shift=0): midpoint circle with horizontal spans.
 * They are cross-checked in tests/ against definitions that do not depend on the trace
 * (scipy.ndimage labelling / hole filling, quad counts) and against hand-derived answers.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int32_t n;        /* number of contours                                   */
    int32_t *offsets; /* n+1 entries, in points                               */
    int32_t *pts;     /* 2*offsets[n] ints: x0,y0,x1,y1,... (image coordinates) */
} orc_contours;

typedef struct { int32_t *d; size_t n, cap; } ivec;
static int ivec_push(ivec *v, int32_t a) {
    if (v->n == v->cap) {
        size_t nc = v->cap ? v->cap * 2 : 1024;
        int32_t *nd = (int32_t *)realloc(v->d, nc * sizeof(int32_t));
        if (!nd) return -1;
        v->d = nd; v->cap = nc;
    }
    v->d[v->n++] = a;
    return 0;
}

static const int CDX[8] = {1, 1, 0, -1, -1, -1, 0, 1};
static const int CDY[8] = {0, -1, -1, -1, 0, 1, 1, 1};

/* Follow one outer border starting at i0 (library routine: border following with nbd = 2,
 * CHAIN_APPROX_SIMPLE keeps a point only where the outgoing direction differs from the
 * incoming one). */
static int fetch_outer(int8_t *i0, int step, int x, int y, ivec *pts)
{
    const int8_t nbd = 2;
    int deltas[16];
    for (int k = 0; k < 8; k++) deltas[k] = deltas[k + 8] = CDY[k] * step + CDX[k];
    int s_end = 4, s = 4;
    int8_t *i1, *i3, *i4 = 0;
    do { s = (s - 1) & 7; i1 = i0 + deltas[s]; } while (*i1 == 0 && s != s_end);
    if (s == s_end) {                       /* isolated pixel */
        *i0 = (int8_t)(nbd | -128);
        if (ivec_push(pts, x) || ivec_push(pts, y)) return -1;
        return 0;
    }
    i3 = i0;
    int prev_s = s ^ 4;
    for (;;) {
        s_end = s;
        for (;;) { i4 = i3 + deltas[++s]; if (*i4 != 0) break; }
        s &= 7;
        if ((unsigned)(s - 1) < (unsigned)s_end) *i3 = (int8_t)(nbd | -128);
        else if (*i3 == 1) *i3 = nbd;
        if (s != prev_s) { if (ivec_push(pts, x) || ivec_push(pts, y)) return -1; }
        prev_s = s;
        x += CDX[s]; y += CDY[s];
        if (i4 == i0 && i3 == i1) break;
        i3 = i4;
        s = (s + 4) & 7;
    }
    return 0;
}

/* cv.findContours(img, RETR_EXTERNAL, CHAIN_APPROX_SIMPLE); img nonzero = foreground.
 * Reference call site: detection.py:204. */
int orc_find_contours_external(const uint8_t *img, int H, int W, orc_contours *out)
{
    const int step = W + 2;
    int8_t *buf = (int8_t *)calloc((size_t)(H + 2) * step, 1);
    if (!buf) return -1;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) buf[(size_t)(y + 1) * step + x + 1] = img[(size_t)y * W + x] != 0;
    ivec pts = {0, 0, 0}, offs = {0, 0, 0};
    ivec_push(&offs, 0);
    const int width = W + 1, height = H + 1;      /* scanner limits on the padded frame */
    for (int y = 1; y < height; y++) {
        int8_t *row = buf + (size_t)y * step;
        int prev = 0, lnbd_x = 0;
        for (int x = 1; x < width; x++) {
            int p = 0;
            for (; x < width && (p = row[x]) == prev; x++) ;
            if (x >= width) break;
            int is_hole = 0;
            if (!(prev == 0 && p == 1)) {
                if (p != 0 || prev < 1) goto resume_scan;
                if (prev & -2) lnbd_x = x - 1;
                is_hole = 1;
            }
            if (is_hole || row[lnbd_x] > 0) goto resume_scan;   /* RETR_EXTERNAL */
            lnbd_x = x;
            if (fetch_outer(row + x, step, x - 1, y - 1, &pts)) { free(buf); free(pts.d); free(offs.d); return -1; }
            ivec_push(&offs, (int32_t)(pts.n / 2));
            p = row[x];
        resume_scan:
            prev = p;
            if (prev & -2) lnbd_x = x;
        }
    }
    free(buf);
    /* reverse discovery order (each new contour is linked at the head of the list) */
    int n = (int)offs.n - 1;
    out->n = n;
    out->offsets = (int32_t *)malloc((size_t)(n + 1) * sizeof(int32_t));
    out->pts = (int32_t *)malloc((pts.n ? pts.n : 1) * sizeof(int32_t));
    int32_t o = 0;
    for (int k = 0; k < n; k++) {
        int src = n - 1 - k;
        int32_t a = offs.d[src], b = offs.d[src + 1];
        out->offsets[k] = o;
        memcpy(out->pts + 2 * (size_t)o, pts.d + 2 * (size_t)a, (size_t)(b - a) * 2 * sizeof(int32_t));
        o += b - a;
    }
    out->offsets[n] = o;
    free(pts.d); free(offs.d);
    return 0;
}

void orc_free_contours(orc_contours *c) { free(c->offsets); free(c->pts); c->offsets = 0; c->pts = 0; c->n = 0; }

/* cv.contourArea(cnt) (detection.py:208) */
double orc_contour_area(const int32_t *pts, int n)
{
    if (n == 0) return 0.0;
    double a00 = 0;
    float px = (float)pts[2 * (n - 1)], py = (float)pts[2 * (n - 1) + 1];
    for (int i = 0; i < n; i++) {
        float x = (float)pts[2 * i], y = (float)pts[2 * i + 1];
        a00 += (double)px * y - (double)py * x;
        px = x; py = y;
    }
    a00 *= 0.5;
    return a00 < 0 ? -a00 : a00;
}

/* cv.boundingRect(cnt) (detection.py:212) -> x,y,w,h */
void orc_bounding_rect(const int32_t *pts, int n, int32_t *xywh)
{
    int x0 = pts[0], x1 = pts[0], y0 = pts[1], y1 = pts[1];
    for (int i = 1; i < n; i++) {
        int x = pts[2 * i], y = pts[2 * i + 1];
        if (x < x0) x0 = x; if (x > x1) x1 = x;
        if (y < y0) y0 = y; if (y > y1) y1 = y;
    }
    xywh[0] = x0; xywh[1] = y0; xywh[2] = x1 - x0 + 1; xywh[3] = y1 - y0 + 1;
}

static int cmp_yx(const void *a, const void *b)
{
    const int32_t *p = (const int32_t *)a, *q = (const int32_t *)b;
    if (p[0] != q[0]) return p[0] < q[0] ? -1 : 1;
    return p[1] < q[1] ? -1 : (p[1] > q[1]);
}

/* cv.drawContours(mask, [cnt], -1, value, thickness=cv.FILLED) (detection.py:245) */
int orc_draw_contour_filled(uint8_t *mask, int H, int W, const int32_t *pts, int n, uint8_t value)
{
    if (n <= 0) return 0;
    /* outline: consecutive vertices of a traced border are joined by exact 8-direction runs */
    for (int i = 0; i < n; i++) {
        int x0 = pts[2 * ((i + n - 1) % n)], y0 = pts[2 * ((i + n - 1) % n) + 1];
        int x1 = pts[2 * i], y1 = pts[2 * i + 1];
        int sx = (x1 > x0) - (x1 < x0), sy = (y1 > y0) - (y1 < y0);
        int x = x0, y = y0;
        for (;;) {
            if (x >= 0 && x < W && y >= 0 && y < H) mask[(size_t)y * W + x] = value;
            if (x == x1 && y == y1) break;
            x += sx; y += sy;
        }
    }
    /* even-odd scanline fill; edge active for y0 <= y < y1 */
    ivec cr = {0, 0, 0};
    int nedges = 0;
    for (int i = 0; i < n; i++) {
        int xa = pts[2 * ((i + n - 1) % n)], ya = pts[2 * ((i + n - 1) % n) + 1];
        int xb = pts[2 * i], yb = pts[2 * i + 1];
        if (ya == yb) continue;
        nedges++;
        if (ya > yb) { int t = xa; xa = xb; xb = t; t = ya; ya = yb; yb = t; }
        int dx = (xb > xa) - (xb < xa);
        for (int y = ya; y < yb; y++) { ivec_push(&cr, y); ivec_push(&cr, xa + dx * (y - ya)); }
    }
    if (nedges >= 2) {
        qsort(cr.d, cr.n / 2, 2 * sizeof(int32_t), cmp_yx);
        for (size_t k = 0; k + 1 < cr.n / 2; k += 2) {
            int y = cr.d[2 * k], xa = cr.d[2 * k + 1], xb = cr.d[2 * k + 3];
            if (y < 0 || y >= H) continue;
            if (xa < 0) xa = 0;
            if (xb >= W) xb = W - 1;
            for (int x = xa; x <= xb; x++) mask[(size_t)y * W + x] = value;
        }
    }
    free(cr.d);
    return 0;
}

/* cv.circle(mask, (cx,cy), radius, value, -1) (detection.py:320-326, segmenter.py:58) */
void orc_circle_filled(uint8_t *img, int H, int W, int cx, int cy, int radius, uint8_t value)
{
    int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
    while (dx >= dy) {
        int ys[4] = {cy - dy, cy + dy, cy - dx, cy + dx};
        int xa[4] = {cx - dx, cx - dx, cx - dy, cx - dy};
        int xb[4] = {cx + dx, cx + dx, cx + dy, cx + dy};
        for (int k = 0; k < 4; k++) {
            int y = ys[k];
            if (y < 0 || y >= H) continue;
            int a = xa[k] < 0 ? 0 : xa[k], b = xb[k] >= W ? W - 1 : xb[k];
            for (int x = a; x <= b; x++) img[(size_t)y * W + x] = value;
        }
        dy++;
        err += plus;
        plus += 2;
        int mask = (err <= 0) - 1;
        err -= minus & mask;
        dx += mask;
        minus -= mask & 2;
    }
}

/* Label map as SURVEY.md section 8(a) defines it (the reference has none): k >= 1 on every
 * foreground pixel of the 8-connected component traced by contour k-1, 0 elsewhere
 * (background and components nested in holes). */
int orc_label_map(const uint8_t *thresh, int H, int W, const orc_contours *c, int32_t *labels)
{
    memset(labels, 0, (size_t)H * W * sizeof(int32_t));
    int32_t *stack = (int32_t *)malloc((size_t)H * W * sizeof(int32_t));
    if (!stack) return -1;
    for (int k = 0; k < c->n; k++) {
        int x = c->pts[2 * (size_t)c->offsets[k]], y = c->pts[2 * (size_t)c->offsets[k] + 1];
        size_t sp = 0;
        labels[(size_t)y * W + x] = k + 1;
        stack[sp++] = y * W + x;
        while (sp) {
            int32_t p = stack[--sp];
            int py = p / W, px = p % W;
            for (int d = 0; d < 8; d++) {
                int qx = px + CDX[d], qy = py + CDY[d];
                if (qx < 0 || qy < 0 || qx >= W || qy >= H) continue;
                size_t q = (size_t)qy * W + qx;
                if (thresh[q] && !labels[q]) { labels[q] = k + 1; stack[sp++] = (int32_t)q; }
            }
        }
    }
    free(stack);
    return 0;
}
