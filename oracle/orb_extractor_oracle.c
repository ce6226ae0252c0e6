/*
 * ORACLE (test infrastructure only; PARITY UNPINNED, see orb_oracle.h).
 *
 * CPU restatement of ORBextractor (reference src/ORBextractor.cc) with the OpenCV primitives it
 * calls (cv::FAST, cv::resize, cv::GaussianBlur, cv::copyMakeBorder, cv::fastAtan2, cvRound)
 * restated from their published plain-C++ (OpenCV 3.2 era, non-IPP) behaviour; OpenCV itself is
 * an un-vendored, unpinned dependency (reference CMakeLists.txt:33-39).
 *
 * Build with -ffp-contract=off (no FMA contraction in the float code).
 */
#include "orb_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define PATCH_SIZE 31      /* src/ORBextractor.cc:72 */
#define HALF_PATCH_SIZE 15 /* :73 */
#define EDGE_THRESHOLD 19  /* :74 */
#define MAX_LEVELS 16

static const int8_t bit_pattern_31[256 * 4] = {
#include "orb_pattern.inc"
};

/* ------------------------------------------------------------------ cvRound / cvFloor / cvCeil */
int orc_cv_round_f(float v) { return (int)lrintf(v); } /* round-half-even (SSE cvtss2si) */
static int cv_round_d(double v) { return (int)lrint(v); }
static int cv_floor_d(double v) { return (int)floor(v); }
static int cv_ceil_d(double v) { return (int)ceil(v); }

/* ------------------------------------------------------------------ cv::fastAtan2 (degrees) */
float orc_fast_atan2(float y, float x)
{
    static const float p1 = 0.9997878412794807f * (float)(180 / 3.1415926535897932384626433832795);
    static const float p3 = -0.3258083974640975f * (float)(180 / 3.1415926535897932384626433832795);
    static const float p5 = 0.1555786518463281f * (float)(180 / 3.1415926535897932384626433832795);
    static const float p7 = -0.04432655554792128f * (float)(180 / 3.1415926535897932384626433832795);
    const float eps = (float)2.2204460492503131e-16; /* (float)DBL_EPSILON */
    float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + eps);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + eps);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

/* ------------------------------------------------------------------ cv::copyMakeBorder REFLECT_101 */
static int reflect101(int p, int n)
{
    /* cv::borderInterpolate(p, n, BORDER_REFLECT_101) */
    if ((unsigned)p < (unsigned)n) return p;
    if (n == 1) return 0;
    do {
        if (p < 0)
            p = -p;
        else
            p = 2 * (n - 1) - p;
    } while ((unsigned)p >= (unsigned)n);
    return p;
}

void orc_copy_make_border_reflect101(const uint8_t *src, int w, int h, int sstride, uint8_t *dst,
                                     int border, int dstride)
{
    for (int y = -border; y < h + border; ++y) {
        const uint8_t *srow = src + (size_t)reflect101(y, h) * sstride;
        uint8_t *drow = dst + (size_t)(y + border) * dstride;
        for (int x = -border; x < w + border; ++x) drow[x + border] = srow[reflect101(x, w)];
    }
}

/* ------------------------------------------------------------------ cv::resize INTER_LINEAR 8UC1
 * fixed point: INTER_RESIZE_COEF_BITS = 11; HResizeLinear<uchar,int,short>, VResizeLinear with
 * FixedPtCast<int,uchar,22>, written in the (b*(S>>4))>>16 form the library uses. */
static short sat_short_round(float v)
{
    int i = (int)lrintf(v);
    if (i > 32767) i = 32767;
    if (i < -32768) i = -32768;
    return (short)i;
}

void orc_resize_linear_u8(const uint8_t *src, int sw, int sh, int sstride, uint8_t *dst, int dw,
                          int dh, int dstride)
{
    double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    int *xofs = (int *)malloc(sizeof(int) * dw);
    short *ialpha = (short *)malloc(sizeof(short) * 2 * dw);
    int *yofs = (int *)malloc(sizeof(int) * dh);
    short *ibeta = (short *)malloc(sizeof(short) * 2 * dh);
    int *row0 = (int *)malloc(sizeof(int) * dw), *row1 = (int *)malloc(sizeof(int) * dw);

    for (int dx = 0; dx < dw; ++dx) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor_d(fx);
        fx -= sx;
        if (sx < 0) {
            fx = 0;
            sx = 0;
        }
        if (sx >= sw - 1) {
            fx = 0;
            sx = sw - 1;
        }
        xofs[dx] = sx;
        ialpha[2 * dx] = sat_short_round((1.f - fx) * 2048);
        ialpha[2 * dx + 1] = sat_short_round(fx * 2048);
    }
    for (int dy = 0; dy < dh; ++dy) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor_d(fy);
        fy -= sy;
        yofs[dy] = sy;
        ibeta[2 * dy] = sat_short_round((1.f - fy) * 2048);
        ibeta[2 * dy + 1] = sat_short_round(fy * 2048);
    }
    for (int dy = 0; dy < dh; ++dy) {
        int sy0 = yofs[dy], sy1 = yofs[dy] + 1;
        if (sy0 < 0) sy0 = 0;
        if (sy0 > sh - 1) sy0 = sh - 1;
        if (sy1 < 0) sy1 = 0;
        if (sy1 > sh - 1) sy1 = sh - 1;
        const uint8_t *S0 = src + (size_t)sy0 * sstride, *S1 = src + (size_t)sy1 * sstride;
        for (int dx = 0; dx < dw; ++dx) {
            int sx = xofs[dx];
            int sx1 = sx + 1 < sw ? sx + 1 : sx; /* coefficient is 0 there */
            row0[dx] = S0[sx] * ialpha[2 * dx] + S0[sx1] * ialpha[2 * dx + 1];
            row1[dx] = S1[sx] * ialpha[2 * dx] + S1[sx1] * ialpha[2 * dx + 1];
        }
        short b0 = ibeta[2 * dy], b1 = ibeta[2 * dy + 1];
        uint8_t *D = dst + (size_t)dy * dstride;
        for (int dx = 0; dx < dw; ++dx) {
            int v = (((b0 * (row0[dx] >> 4)) >> 16) + ((b1 * (row1[dx] >> 4)) >> 16) + 2) >> 2;
            D[dx] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
    }
    free(xofs);
    free(ialpha);
    free(yofs);
    free(ibeta);
    free(row0);
    free(row1);
}

/* ------------------------------------------------------------------ cv::GaussianBlur 7x7 sigma=2 8U
 * getGaussianKernel(7, 2, CV_32F) -> createSeparableLinearFilter 8U symmetric path: kernel
 * converted to int with 8 fractional bits, int32 row pass, column pass (sum + 2^15) >> 16. */
static void gaussian_kernel7_q8(int k[7])
{
    const int n = 7;
    const double sigmaX = 2.0;
    double scale2X = -0.5 / (sigmaX * sigmaX);
    float cf[7];
    double sum = 0;
    for (int i = 0; i < n; ++i) {
        double x = i - (n - 1) * 0.5;
        double t = exp(scale2X * x * x);
        cf[i] = (float)t;
        sum += cf[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < n; ++i) {
        cf[i] = (float)(cf[i] * sum);
        k[i] = cv_round_d((double)cf[i] * 256.0);
    }
}

void orc_gaussian_blur7_u8(const uint8_t *src, int w, int h, int sstride, uint8_t *dst, int dstride)
{
    int k[7];
    gaussian_kernel7_q8(k);
    int *tmp = (int *)malloc(sizeof(int) * (size_t)w * h);
    for (int y = 0; y < h; ++y) {
        const uint8_t *s = src + (size_t)y * sstride;
        for (int x = 0; x < w; ++x) {
            int acc = 0;
            for (int t = -3; t <= 3; ++t) acc += k[t + 3] * s[reflect101(x + t, w)];
            tmp[(size_t)y * w + x] = acc;
        }
    }
    for (int y = 0; y < h; ++y) {
        uint8_t *d = dst + (size_t)y * dstride;
        for (int x = 0; x < w; ++x) {
            int acc = 0;
            for (int t = -3; t <= 3; ++t) acc += k[t + 3] * tmp[(size_t)reflect101(y + t, h) * w + x];
            int v = (acc + (1 << 15)) >> 16;
            d[x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
    }
    free(tmp);
}

/* ------------------------------------------------------------------ cv::FAST TYPE_9_16 */
static const int fast_off[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},  {3, 0},  {3, -1},
                                    {2, -2}, {1, -3},  {0, -3},  {-1, -3}, {-2, -2}, {-3, -1},
                                    {-3, 0}, {-3, 1},  {-2, 2},  {-1, 3}};

static void fast_make_offsets(int pixel[25], int stride)
{
    for (int k = 0; k < 16; ++k) pixel[k] = fast_off[k][0] + fast_off[k][1] * stride;
    for (int k = 16; k < 25; ++k) pixel[k] = pixel[k - 16];
}

/* cornerScore<16> of OpenCV's fast_score.cpp */
static int fast_corner_score16(const uint8_t *ptr, const int pixel[25], int threshold)
{
    const int K = 8, N = K * 3 + 1;
    int k, v = ptr[0];
    short d[25];
    for (k = 0; k < N; k++) d[k] = (short)(v - ptr[pixel[k]]);

    int a0 = threshold;
    for (k = 0; k < 16; k += 2) {
        int a = d[k + 1] < d[k + 2] ? d[k + 1] : d[k + 2];
        a = a < d[k + 3] ? a : d[k + 3];
        if (a <= a0) continue;
        for (int j = 4; j <= 8; ++j) a = a < d[k + j] ? a : d[k + j];
        int m0 = a < d[k] ? a : d[k];
        int m1 = a < d[k + 9] ? a : d[k + 9];
        a0 = a0 > m0 ? a0 : m0;
        a0 = a0 > m1 ? a0 : m1;
    }
    int b0 = -a0;
    for (k = 0; k < 16; k += 2) {
        int b = d[k + 1] > d[k + 2] ? d[k + 1] : d[k + 2];
        for (int j = 3; j <= 5; ++j) b = b > d[k + j] ? b : d[k + j];
        if (b >= b0) continue;
        for (int j = 6; j <= 8; ++j) b = b > d[k + j] ? b : d[k + j];
        int m0 = b > d[k] ? b : d[k];
        int m1 = b > d[k + 9] ? b : d[k + 9];
        b0 = b0 < m0 ? b0 : m0;
        b0 = b0 < m1 ? b0 : m1;
    }
    return -b0 - 1;
}

int orc_fast_corner_score(const uint8_t *center, int stride, int threshold)
{
    int pixel[25];
    fast_make_offsets(pixel, stride);
    return fast_corner_score16(center, pixel, threshold);
}

/* FAST_t<16>(img, keypoints, threshold, nonmax_suppression=true) */
int orc_fast9_16(const uint8_t *img, int w, int h, int stride, int threshold, int16_t *xs,
                 int16_t *ys, uint8_t *score, int cap)
{
    const int K = 8, N = 25;
    int pixel[25];
    fast_make_offsets(pixel, stride);
    if (threshold < 0) threshold = 0;
    if (threshold > 255) threshold = 255;
    if (w < 7 || h < 7) return 0;
    uint8_t *sc = (uint8_t *)calloc((size_t)w * h, 1);
    uint8_t threshold_tab[512];
    for (int i = -255; i <= 255; i++)
        threshold_tab[i + 255] = (uint8_t)(i < -threshold ? 1 : i > threshold ? 2 : 0);
    for (int i = 3; i < h - 3; ++i) {
        const uint8_t *ptr = img + (size_t)i * stride + 3;
        for (int j = 3; j < w - 3; ++j, ++ptr) {
            int v = ptr[0];
            const uint8_t *tab = &threshold_tab[0] - v + 255;
            int d = tab[ptr[pixel[0]]] | tab[ptr[pixel[8]]];
            if (d == 0) continue;
            d &= tab[ptr[pixel[2]]] | tab[ptr[pixel[10]]];
            d &= tab[ptr[pixel[4]]] | tab[ptr[pixel[12]]];
            d &= tab[ptr[pixel[6]]] | tab[ptr[pixel[14]]];
            if (d == 0) continue;
            d &= tab[ptr[pixel[1]]] | tab[ptr[pixel[9]]];
            d &= tab[ptr[pixel[3]]] | tab[ptr[pixel[11]]];
            d &= tab[ptr[pixel[5]]] | tab[ptr[pixel[13]]];
            d &= tab[ptr[pixel[7]]] | tab[ptr[pixel[15]]];
            int is_corner = 0;
            if (d & 1) { /* dark arc: x < v - threshold */
                int vt = v - threshold, count = 0;
                for (int k = 0; k < N; ++k) {
                    int x = ptr[pixel[k]];
                    if (x < vt) {
                        if (++count > K) {
                            is_corner = 1;
                            break;
                        }
                    } else
                        count = 0;
                }
            }
            if (!is_corner && (d & 2)) {
                int vt = v + threshold, count = 0;
                for (int k = 0; k < N; ++k) {
                    int x = ptr[pixel[k]];
                    if (x > vt) {
                        if (++count > K) {
                            is_corner = 1;
                            break;
                        }
                    } else
                        count = 0;
                }
            }
            if (is_corner) sc[(size_t)i * w + j] = (uint8_t)fast_corner_score16(ptr, pixel, threshold);
        }
    }
    int n = 0;
    for (int i = 3; i < h - 3; ++i) {
        for (int j = 3; j < w - 3; ++j) {
            int s = sc[(size_t)i * w + j];
            if (!s) continue; /* not in cornerpos */
            const uint8_t *p = sc + (size_t)i * w + j;
            if (s > p[-1] && s > p[1] && s > p[-w - 1] && s > p[-w] && s > p[-w + 1] &&
                s > p[w - 1] && s > p[w] && s > p[w + 1]) {
                if (n < cap) {
                    xs[n] = (int16_t)j;
                    ys[n] = (int16_t)i;
                    score[n] = (uint8_t)s;
                }
                ++n;
            }
        }
    }
    free(sc);
    return n;
}

/* ------------------------------------------------------------------ DistributeOctTree
 * src/ORBextractor.cc:481-763.  std::list<ExtractorNode> is emulated by an index-linked arena.
 * Pointer-order tie-break of the sort at :684 is replaced by creation order (node index),
 * later-created = larger (parity convention 1). */
typedef struct {
    int ULx, ULy, URx, URy, BLx, BLy, BRx, BRy;
    int *keys;
    int nkeys;
    int no_more;
    int prev, next;
} onode_t;

typedef struct {
    onode_t *nodes;
    int n_alloc, cap;
    int head, tail, size;
} olist_t;

static int ol_new(olist_t *L)
{
    if (L->n_alloc == L->cap) {
        L->cap = L->cap ? L->cap * 2 : 256;
        L->nodes = (onode_t *)realloc(L->nodes, sizeof(onode_t) * L->cap);
    }
    onode_t *n = &L->nodes[L->n_alloc];
    memset(n, 0, sizeof(*n));
    n->prev = n->next = -1;
    return L->n_alloc++;
}
static void ol_push_back(olist_t *L, int id)
{
    onode_t *n = &L->nodes[id];
    n->prev = L->tail;
    n->next = -1;
    if (L->tail >= 0)
        L->nodes[L->tail].next = id;
    else
        L->head = id;
    L->tail = id;
    L->size++;
}
static void ol_push_front(olist_t *L, int id)
{
    onode_t *n = &L->nodes[id];
    n->next = L->head;
    n->prev = -1;
    if (L->head >= 0)
        L->nodes[L->head].prev = id;
    else
        L->tail = id;
    L->head = id;
    L->size++;
}
/* returns next */
static int ol_erase(olist_t *L, int id)
{
    onode_t *n = &L->nodes[id];
    int nx = n->next;
    if (n->prev >= 0)
        L->nodes[n->prev].next = n->next;
    else
        L->head = n->next;
    if (n->next >= 0)
        L->nodes[n->next].prev = n->prev;
    else
        L->tail = n->prev;
    L->size--;
    free(n->keys);
    n->keys = NULL;
    return nx;
}

/* ExtractorNode::DivideNode :481-537; children are arena ids c[0..3] (not yet linked) */
static void divide_node(olist_t *L, int id, int c[4], const float *xs, const float *ys)
{
    for (int i = 0; i < 4; ++i) c[i] = ol_new(L);
    onode_t *p = &L->nodes[id];
    const int halfX = (int)ceilf((float)(p->URx - p->ULx) / 2);
    const int halfY = (int)ceilf((float)(p->BRy - p->ULy) / 2);
    onode_t *n1 = &L->nodes[c[0]], *n2 = &L->nodes[c[1]], *n3 = &L->nodes[c[2]], *n4 = &L->nodes[c[3]];
    n1->ULx = p->ULx; n1->ULy = p->ULy;
    n1->URx = p->ULx + halfX; n1->URy = p->ULy;
    n1->BLx = p->ULx; n1->BLy = p->ULy + halfY;
    n1->BRx = p->ULx + halfX; n1->BRy = p->ULy + halfY;
    n2->ULx = n1->URx; n2->ULy = n1->URy;
    n2->URx = p->URx; n2->URy = p->URy;
    n2->BLx = n1->BRx; n2->BLy = n1->BRy;
    n2->BRx = p->URx; n2->BRy = p->ULy + halfY;
    n3->ULx = n1->BLx; n3->ULy = n1->BLy;
    n3->URx = n1->BRx; n3->URy = n1->BRy;
    n3->BLx = p->BLx; n3->BLy = p->BLy;
    n3->BRx = n1->BRx; n3->BRy = p->BLy;
    n4->ULx = n3->URx; n4->ULy = n3->URy;
    n4->URx = n2->BRx; n4->URy = n2->BRy;
    n4->BLx = n3->BRx; n4->BLy = n3->BRy;
    n4->BRx = p->BRx; n4->BRy = p->BRy;
    for (int i = 0; i < 4; ++i) L->nodes[c[i]].keys = (int *)malloc(sizeof(int) * (p->nkeys ? p->nkeys : 1));
    for (int i = 0; i < p->nkeys; ++i) {
        int k = p->keys[i];
        float x = xs[k], y = ys[k];
        onode_t *t;
        if (x < (float)n1->URx) {
            if (y < (float)n1->BRy) t = n1; else t = n3;
        } else if (y < (float)n1->BRy)
            t = n2;
        else
            t = n4;
        t->keys[t->nkeys++] = k;
    }
    for (int i = 0; i < 4; ++i)
        if (L->nodes[c[i]].nkeys == 1) L->nodes[c[i]].no_more = 1;
}

typedef struct { int size, node; } spair_t;
/* orc_set_tiebreak_mode(1): nodes of equal size in the OPPOSITE order (the reference's order among them is that of their heap
 * addresses, i.e. unspecified: the two extremes bracket what a reference binary can do) -- for measuring the convention's effect
 * (tools/convention_effects.py); the parity tests never set it */
static int g_tiebreak_reverse = 0;
void orc_set_tiebreak_mode(int reverse) { g_tiebreak_reverse = reverse; }
static int spair_cmp(const void *a, const void *b)
{
    const spair_t *x = (const spair_t *)a, *y = (const spair_t *)b;
    if (x->size != y->size) return x->size < y->size ? -1 : 1;
    if (g_tiebreak_reverse) return x->node > y->node ? -1 : x->node < y->node ? 1 : 0;
    return x->node < y->node ? -1 : x->node > y->node ? 1 : 0;
}

int orc_distribute_octree(const float *xs, const float *ys, const float *resp, int n, int minX,
                          int maxX, int minY, int maxY, int N, int *out_idx, int cap)
{
    olist_t L;
    memset(&L, 0, sizeof(L));
    L.head = L.tail = -1;
    const int nIni = (int)roundf((float)(maxX - minX) / (maxY - minY));
    if (nIni < 1) return -1; /* the reference divides by zero here (:545-547, levels more than twice as tall as wide) */
    const float hX = (float)(maxX - minX) / nIni;
    int *ini = (int *)malloc(sizeof(int) * (nIni > 0 ? nIni : 1));
    for (int i = 0; i < nIni; ++i) {
        int id = ol_new(&L);
        onode_t *ni = &L.nodes[id];
        ni->ULx = (int)(hX * (float)i); ni->ULy = 0;
        ni->URx = (int)(hX * (float)(i + 1)); ni->URy = 0;
        ni->BLx = ni->ULx; ni->BLy = maxY - minY;
        ni->BRx = ni->URx; ni->BRy = maxY - minY;
        ni->keys = (int *)malloc(sizeof(int) * (n ? n : 1));
        ol_push_back(&L, id);
        ini[i] = id;
    }
    for (int i = 0; i < n; ++i) {
        int r = (int)(xs[i] / hX);
        onode_t *nd = &L.nodes[ini[r]];
        nd->keys[nd->nkeys++] = i;
    }
    for (int lit = L.head; lit >= 0;) {
        onode_t *nd = &L.nodes[lit];
        if (nd->nkeys == 1) {
            nd->no_more = 1;
            lit = nd->next;
        } else if (nd->nkeys == 0)
            lit = ol_erase(&L, lit);
        else
            lit = nd->next;
    }
    int finish = 0;
    spair_t *vsz = NULL, *vprev = NULL;
    int nvsz = 0, capv = 0;
#define VSZ_PUSH(sz_, nd_)                                                    \
    do {                                                                      \
        if (nvsz == capv) {                                                   \
            capv = capv ? capv * 2 : 256;                                     \
            vsz = (spair_t *)realloc(vsz, sizeof(spair_t) * capv);            \
        }                                                                     \
        vsz[nvsz].size = (sz_);                                               \
        vsz[nvsz].node = (nd_);                                               \
        nvsz++;                                                               \
    } while (0)

    while (!finish) {
        int prevSize = L.size;
        int lit = L.head;
        int nToExpand = 0;
        nvsz = 0;
        while (lit >= 0) {
            if (L.nodes[lit].no_more) {
                lit = L.nodes[lit].next;
                continue;
            }
            int c[4];
            divide_node(&L, lit, c, xs, ys);
            for (int i = 0; i < 4; ++i) {
                if (L.nodes[c[i]].nkeys > 0) {
                    ol_push_front(&L, c[i]);
                    if (L.nodes[c[i]].nkeys > 1) {
                        nToExpand++;
                        VSZ_PUSH(L.nodes[c[i]].nkeys, c[i]);
                    }
                } else {
                    free(L.nodes[c[i]].keys);
                    L.nodes[c[i]].keys = NULL;
                }
            }
            lit = ol_erase(&L, lit);
        }
        if (L.size >= N || L.size == prevSize) {
            finish = 1;
        } else if (L.size + nToExpand * 3 > N) {
            while (!finish) {
                prevSize = L.size;
                int nprev = nvsz;
                vprev = (spair_t *)realloc(vprev, sizeof(spair_t) * (nprev ? nprev : 1));
                memcpy(vprev, vsz, sizeof(spair_t) * nprev);
                nvsz = 0;
                qsort(vprev, nprev, sizeof(spair_t), spair_cmp);
                for (int j = nprev - 1; j >= 0; --j) {
                    int c[4];
                    divide_node(&L, vprev[j].node, c, xs, ys);
                    for (int i = 0; i < 4; ++i) {
                        if (L.nodes[c[i]].nkeys > 0) {
                            ol_push_front(&L, c[i]);
                            if (L.nodes[c[i]].nkeys > 1) VSZ_PUSH(L.nodes[c[i]].nkeys, c[i]);
                        } else {
                            free(L.nodes[c[i]].keys);
                            L.nodes[c[i]].keys = NULL;
                        }
                    }
                    ol_erase(&L, vprev[j].node);
                    if (L.size >= N) break;
                }
                if (L.size >= N || L.size == prevSize) finish = 1;
            }
        }
    }
#undef VSZ_PUSH
    int nout = 0;
    for (int lit = L.head; lit >= 0; lit = L.nodes[lit].next) {
        onode_t *nd = &L.nodes[lit];
        int best = nd->keys[0];
        float maxResponse = resp[best];
        for (int k = 1; k < nd->nkeys; ++k) {
            if (resp[nd->keys[k]] > maxResponse) {
                best = nd->keys[k];
                maxResponse = resp[best];
            }
        }
        if (nout < cap) out_idx[nout] = best;
        nout++;
    }
    for (int i = 0; i < L.n_alloc; ++i) free(L.nodes[i].keys);
    free(L.nodes);
    free(vsz);
    free(vprev);
    free(ini);
    return nout;
}

/* ------------------------------------------------------------------ extractor object */
struct orc_extractor {
    int nfeatures, nlevels, iniTh, minTh;
    float scaleFactor;
    float mvScaleFactor[MAX_LEVELS], mvInvScaleFactor[MAX_LEVELS];
    float mvLevelSigma2[MAX_LEVELS], mvInvLevelSigma2[MAX_LEVELS];
    int mnFeaturesPerLevel[MAX_LEVELS];
    int umax[HALF_PATCH_SIZE + 1];
    /* per-call state */
    int lw[MAX_LEVELS], lh[MAX_LEVELS], lpitch[MAX_LEVELS];
    uint8_t *plane[MAX_LEVELS];   /* bordered buffer */
    uint8_t *blurred[MAX_LEVELS]; /* w x h contiguous */
    int ncand[MAX_LEVELS];
    int16_t *cx[MAX_LEVELS], *cy[MAX_LEVELS];
    uint8_t *cs[MAX_LEVELS];
    int nkeys[MAX_LEVELS];
    orc_keypoint_t *keys[MAX_LEVELS];
};

orc_extractor_t *orc_extractor_create(int nfeatures, float scaleFactor, int nlevels, int iniTh,
                                      int minTh)
{
    if (nlevels < 1 || nlevels > MAX_LEVELS) return NULL;
    orc_extractor_t *e = (orc_extractor_t *)calloc(1, sizeof(*e));
    e->nfeatures = nfeatures;
    e->nlevels = nlevels;
    e->iniTh = iniTh;
    e->minTh = minTh;
    e->scaleFactor = scaleFactor;
    /* :415-432 */
    e->mvScaleFactor[0] = 1.0f;
    e->mvLevelSigma2[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) {
        e->mvScaleFactor[i] = e->mvScaleFactor[i - 1] * scaleFactor;
        e->mvLevelSigma2[i] = e->mvScaleFactor[i] * e->mvScaleFactor[i];
    }
    for (int i = 0; i < nlevels; i++) {
        e->mvInvScaleFactor[i] = 1.0f / e->mvScaleFactor[i];
        e->mvInvLevelSigma2[i] = 1.0f / e->mvLevelSigma2[i];
    }
    /* :436-448 */
    float factor = 1.0f / scaleFactor;
    float nDesiredFeaturesPerScale =
        nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
    int sumFeatures = 0;
    for (int level = 0; level < nlevels - 1; level++) {
        e->mnFeaturesPerLevel[level] = orc_cv_round_f(nDesiredFeaturesPerScale);
        sumFeatures += e->mnFeaturesPerLevel[level];
        nDesiredFeaturesPerScale *= factor;
    }
    e->mnFeaturesPerLevel[nlevels - 1] = nfeatures - sumFeatures > 0 ? nfeatures - sumFeatures : 0;
    /* :454-470 */
    int v, v0, vmax = cv_floor_d(HALF_PATCH_SIZE * sqrtf(2.f) / 2 + 1);
    int vmin = cv_ceil_d(HALF_PATCH_SIZE * sqrtf(2.f) / 2);
    const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
    for (v = 0; v <= vmax; ++v) e->umax[v] = cv_round_d(sqrt(hp2 - v * v));
    for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
        while (e->umax[v0] == e->umax[v0 + 1]) ++v0;
        e->umax[v] = v0;
        ++v0;
    }
    return e;
}

static void free_call_state(orc_extractor_t *e)
{
    for (int l = 0; l < MAX_LEVELS; ++l) {
        free(e->plane[l]); e->plane[l] = NULL;
        free(e->blurred[l]); e->blurred[l] = NULL;
        free(e->cx[l]); e->cx[l] = NULL;
        free(e->cy[l]); e->cy[l] = NULL;
        free(e->cs[l]); e->cs[l] = NULL;
        free(e->keys[l]); e->keys[l] = NULL;
        e->ncand[l] = e->nkeys[l] = 0;
    }
}

void orc_extractor_destroy(orc_extractor_t *e)
{
    if (!e) return;
    free_call_state(e);
    free(e);
}

int orc_extractor_levels(const orc_extractor_t *e) { return e->nlevels; }
const float *orc_extractor_scale_factors(const orc_extractor_t *e) { return e->mvScaleFactor; }
const float *orc_extractor_inv_scale_factors(const orc_extractor_t *e) { return e->mvInvScaleFactor; }
const float *orc_extractor_sigma2(const orc_extractor_t *e) { return e->mvLevelSigma2; }
const float *orc_extractor_inv_sigma2(const orc_extractor_t *e) { return e->mvInvLevelSigma2; }
const int *orc_extractor_features_per_level(const orc_extractor_t *e) { return e->mnFeaturesPerLevel; }
const int *orc_extractor_umax(const orc_extractor_t *e) { return e->umax; }

int orc_extractor_level_size(const orc_extractor_t *e, int level, int *w, int *h)
{
    if (level < 0 || level >= e->nlevels || !e->plane[level]) return -1;
    *w = e->lw[level];
    *h = e->lh[level];
    return e->lpitch[level];
}
const uint8_t *orc_extractor_level_plane(const orc_extractor_t *e, int level)
{
    if (level < 0 || level >= e->nlevels || !e->plane[level]) return NULL;
    return e->plane[level] + (size_t)EDGE_THRESHOLD * e->lpitch[level] + EDGE_THRESHOLD;
}
const uint8_t *orc_extractor_level_blurred(const orc_extractor_t *e, int level)
{
    if (level < 0 || level >= e->nlevels) return NULL;
    return e->blurred[level];
}
int orc_extractor_level_candidates(const orc_extractor_t *e, int level, const int16_t **xs,
                                   const int16_t **ys, const uint8_t **score)
{
    if (level < 0 || level >= e->nlevels) return -1;
    *xs = e->cx[level];
    *ys = e->cy[level];
    *score = e->cs[level];
    return e->ncand[level];
}
int orc_extractor_level_nkeys(const orc_extractor_t *e, int level) { return e->nkeys[level]; }

/* ComputePyramid :1107-1132 */
static void compute_pyramid(orc_extractor_t *e, const uint8_t *img, int w, int h, int stride)
{
    for (int level = 0; level < e->nlevels; ++level) {
        float scale = e->mvInvScaleFactor[level];
        int lw = orc_cv_round_f((float)w * scale), lh = orc_cv_round_f((float)h * scale);
        int pitch = lw + EDGE_THRESHOLD * 2;
        e->lw[level] = lw;
        e->lh[level] = lh;
        e->lpitch[level] = pitch;
        e->plane[level] = (uint8_t *)malloc((size_t)pitch * (lh + EDGE_THRESHOLD * 2));
        uint8_t *interior = e->plane[level] + (size_t)EDGE_THRESHOLD * pitch + EDGE_THRESHOLD;
        if (level != 0) {
            const uint8_t *prev = e->plane[level - 1] + (size_t)EDGE_THRESHOLD * e->lpitch[level - 1] + EDGE_THRESHOLD;
            uint8_t *tmp = (uint8_t *)malloc((size_t)lw * lh);
            orc_resize_linear_u8(prev, e->lw[level - 1], e->lh[level - 1], e->lpitch[level - 1], tmp, lw, lh, lw);
            orc_copy_make_border_reflect101(tmp, lw, lh, lw, e->plane[level], EDGE_THRESHOLD, pitch);
            free(tmp);
        } else {
            orc_copy_make_border_reflect101(img, w, h, stride, e->plane[level], EDGE_THRESHOLD, pitch);
        }
        (void)interior;
    }
}

/* IC_Angle :77-104 */
static float ic_angle(const uint8_t *image, int step, float ptx, float pty, const int *u_max)
{
    int m_01 = 0, m_10 = 0;
    const uint8_t *center = image + (size_t)orc_cv_round_f(pty) * step + orc_cv_round_f(ptx);
    for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
    for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
        int v_sum = 0;
        int d = u_max[v];
        for (int u = -d; u <= d; ++u) {
            int val_plus = center[u + v * step], val_minus = center[u - v * step];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return orc_fast_atan2((float)m_01, (float)m_10);
}

/* Steering a = cos(angle), b = sin(angle) (:112-113).  The reference calls libm cosf/sinf through
 * the std:: float overloads; their last bit depends on the glibc version (glibc 2.35 differs from
 * the correctly rounded value for ~2.6% of angles).  Parity convention 3: a and b are
 * the correctly rounded float cosine/sine, obtained from a fixed IEEE-double polynomial (same
 * operation sequence as the device).  orc_set_trig_mode(1) switches to libm for measuring the
 * effect of that choice. */
static int g_trig_libm = 0;
void orc_set_trig_mode(int use_libm) { g_trig_libm = use_libm; }

void orc_sincos_exact(float angle_rad, float *s_out, float *c_out)
{
    const double x = (double)angle_rad;
    const int k = (int)(x * 6.36619772367581382433e-01 + 0.5);
    const double fk = (double)k;
    double r = x - fk * 1.57079632673412561417e+00;
    r = r - fk * 6.07710050650619224932e-11;
    const double z = r * r;
    double ps = 1.58969099521155010221e-10;
    ps = ps * z + -2.50507602534068634195e-08;
    ps = ps * z + 2.75573137070700676789e-06;
    ps = ps * z + -1.98412698298579493134e-04;
    ps = ps * z + 8.33333333332248946124e-03;
    ps = ps * z + -1.66666666666666324348e-01;
    const double sn = r + (r * z) * ps;
    double pc = -1.13596475577881948265e-11;
    pc = pc * z + 2.08757232129817482790e-09;
    pc = pc * z + -2.75573143513906633035e-07;
    pc = pc * z + 2.48015872894767294178e-05;
    pc = pc * z + -1.38888888888741095749e-03;
    pc = pc * z + 4.16666666666666019037e-02;
    const double cs = (1.0 - 0.5 * z) + (z * z) * pc;
    double sv, cv;
    switch (k & 3) {
        case 0: sv = sn; cv = cs; break;
        case 1: sv = cs; cv = -sn; break;
        case 2: sv = -sn; cv = -cs; break;
        default: sv = -cs; cv = sn; break;
    }
    *s_out = (float)sv;
    *c_out = (float)cv;
}

/* computeOrbDescriptor :108-147 */
static void compute_orb_descriptor(const orc_keypoint_t *kpt, const uint8_t *img, int step,
                                   const int8_t *pattern, uint8_t *desc)
{
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    float angle = (float)kpt->angle * factorPI;
    float a, b;
    if (g_trig_libm) {
        a = (float)cosf(angle);
        b = (float)sinf(angle);
    } else
        orc_sincos_exact(angle, &b, &a);
    const uint8_t *center = img + (size_t)orc_cv_round_f(kpt->y) * step + orc_cv_round_f(kpt->x);
#define GET_VALUE(idx)                                                                        \
    center[orc_cv_round_f((float)pattern[2 * (idx)] * b + (float)pattern[2 * (idx) + 1] * a) * step + \
           orc_cv_round_f((float)pattern[2 * (idx)] * a - (float)pattern[2 * (idx) + 1] * b)]
    for (int i = 0; i < 32; ++i, pattern += 32) {
        int val = 0;
        for (int k = 0; k < 8; ++k) {
            int t0 = GET_VALUE(2 * k), t1 = GET_VALUE(2 * k + 1);
            val |= (t0 < t1) << k;
        }
        desc[i] = (uint8_t)val;
    }
#undef GET_VALUE
}

/* ComputeKeyPointsOctTree :765-853 */
static void compute_keypoints_octree(orc_extractor_t *e)
{
    const float W = 30;
    for (int level = 0; level < e->nlevels; ++level) {
        const int cols = e->lw[level], rows = e->lh[level], pitch = e->lpitch[level];
        const uint8_t *roi = e->plane[level] + (size_t)EDGE_THRESHOLD * pitch + EDGE_THRESHOLD;
        const int minBorderX = EDGE_THRESHOLD - 3;
        const int minBorderY = minBorderX;
        const int maxBorderX = cols - EDGE_THRESHOLD + 3;
        const int maxBorderY = rows - EDGE_THRESHOLD + 3;
        const float width = (float)(maxBorderX - minBorderX);
        const float height = (float)(maxBorderY - minBorderY);
        const int nCols = (int)(width / W);
        const int nRows = (int)(height / W);
        const int wCell = (int)ceilf(width / nCols);
        const int hCell = (int)ceilf(height / nRows);

        int capc = 1024, nc = 0;
        int16_t *cx = (int16_t *)malloc(sizeof(int16_t) * capc);
        int16_t *cy = (int16_t *)malloc(sizeof(int16_t) * capc);
        uint8_t *cs = (uint8_t *)malloc(capc);
        int cellcap = (wCell + 6) * (hCell + 6);
        int16_t *tx = (int16_t *)malloc(sizeof(int16_t) * cellcap);
        int16_t *ty = (int16_t *)malloc(sizeof(int16_t) * cellcap);
        uint8_t *ts = (uint8_t *)malloc(cellcap);

        for (int i = 0; i < nRows; i++) {
            const float iniY = (float)(minBorderY + i * hCell);
            float maxY = iniY + hCell + 6;
            if (iniY >= maxBorderY - 3) continue;
            if (maxY > maxBorderY) maxY = (float)maxBorderY;
            for (int j = 0; j < nCols; j++) {
                const float iniX = (float)(minBorderX + j * wCell);
                float maxX = iniX + wCell + 6;
                if (iniX >= maxBorderX - 6) continue;
                if (maxX > maxBorderX) maxX = (float)maxBorderX;
                const uint8_t *sub = roi + (size_t)(int)iniY * pitch + (int)iniX;
                int sw = (int)maxX - (int)iniX, sh = (int)maxY - (int)iniY;
                int n = orc_fast9_16(sub, sw, sh, pitch, e->iniTh, tx, ty, ts, cellcap);
                if (n == 0) n = orc_fast9_16(sub, sw, sh, pitch, e->minTh, tx, ty, ts, cellcap);
                for (int k = 0; k < n; ++k) {
                    if (nc == capc) {
                        capc *= 2;
                        cx = (int16_t *)realloc(cx, sizeof(int16_t) * capc);
                        cy = (int16_t *)realloc(cy, sizeof(int16_t) * capc);
                        cs = (uint8_t *)realloc(cs, capc);
                    }
                    cx[nc] = (int16_t)(tx[k] + j * wCell);
                    cy[nc] = (int16_t)(ty[k] + i * hCell);
                    cs[nc] = ts[k];
                    nc++;
                }
            }
        }
        free(tx); free(ty); free(ts);
        e->cx[level] = cx; e->cy[level] = cy; e->cs[level] = cs; e->ncand[level] = nc;

        float *fx = (float *)malloc(sizeof(float) * (nc ? nc : 1));
        float *fy = (float *)malloc(sizeof(float) * (nc ? nc : 1));
        float *fr = (float *)malloc(sizeof(float) * (nc ? nc : 1));
        for (int k = 0; k < nc; ++k) {
            fx[k] = (float)cx[k];
            fy[k] = (float)cy[k];
            fr[k] = (float)cs[k];
        }
        int capo = nc > 0 ? nc : 1;
        int *sel = (int *)malloc(sizeof(int) * capo);
        int nk = 0;
        if (nc > 0)
            nk = orc_distribute_octree(fx, fy, fr, nc, minBorderX, maxBorderX, minBorderY, maxBorderY,
                                       e->mnFeaturesPerLevel[level], sel, capo);
        const int scaledPatchSize = (int)(PATCH_SIZE * e->mvScaleFactor[level]);
        e->keys[level] = (orc_keypoint_t *)malloc(sizeof(orc_keypoint_t) * (nk ? nk : 1));
        e->nkeys[level] = nk;
        for (int k = 0; k < nk; ++k) {
            orc_keypoint_t *kp = &e->keys[level][k];
            kp->x = fx[sel[k]] + minBorderX;
            kp->y = fy[sel[k]] + minBorderY;
            kp->octave = level;
            kp->size = (float)scaledPatchSize;
            kp->angle = -1;
            kp->response = fr[sel[k]];
            kp->class_id = -1;
        }
        free(fx); free(fy); free(fr); free(sel);
    }
    for (int level = 0; level < e->nlevels; ++level) {
        const uint8_t *roi = e->plane[level] + (size_t)EDGE_THRESHOLD * e->lpitch[level] + EDGE_THRESHOLD;
        for (int k = 0; k < e->nkeys[level]; ++k)
            e->keys[level][k].angle = ic_angle(roi, e->lpitch[level], e->keys[level][k].x, e->keys[level][k].y, e->umax);
    }
}

/* operator() :1043-1105 */
int orc_extractor_extract(orc_extractor_t *e, const uint8_t *img, int w, int h, int stride,
                          orc_keypoint_t *kps, uint8_t *desc, int cap, int *n_out)
{
    if (n_out) *n_out = 0;
    if (!img || w <= 0 || h <= 0) return 0; /* empty image: silent return (:1046) */
    free_call_state(e);
    {
        /* the reference would divide by zero below this size (nCols/nRows == 0, :783-786) */
        float s = e->mvInvScaleFactor[e->nlevels - 1];
        int lw = orc_cv_round_f((float)w * s), lh = orc_cv_round_f((float)h * s);
        if (lw - 2 * EDGE_THRESHOLD + 6 < 30 || lh - 2 * EDGE_THRESHOLD + 6 < 30) return -3;
        /* DistributeOctTree: nIni = round(width / height) == 0 -> the reference divides by zero (:545-547) */
        for (int level = 0; level < e->nlevels; ++level) {
            float sl = e->mvInvScaleFactor[level];
            int wl = orc_cv_round_f((float)w * sl), hl = orc_cv_round_f((float)h * sl);
            if ((int)roundf((float)(wl - 2 * EDGE_THRESHOLD + 6) / (float)(hl - 2 * EDGE_THRESHOLD + 6)) < 1) return -4;
        }
    }
    compute_pyramid(e, img, w, h, stride);
    compute_keypoints_octree(e);
    int nkeypoints = 0;
    for (int level = 0; level < e->nlevels; ++level) nkeypoints += e->nkeys[level];
    if (n_out) *n_out = nkeypoints;
    int offset = 0;
    int status = 0;
    for (int level = 0; level < e->nlevels; ++level) {
        int lw = e->lw[level], lh = e->lh[level];
        const uint8_t *roi = e->plane[level] + (size_t)EDGE_THRESHOLD * e->lpitch[level] + EDGE_THRESHOLD;
        e->blurred[level] = (uint8_t *)malloc((size_t)lw * lh);
        orc_gaussian_blur7_u8(roi, lw, lh, e->lpitch[level], e->blurred[level], lw);
        int nk = e->nkeys[level];
        for (int k = 0; k < nk; ++k) {
            orc_keypoint_t kp = e->keys[level][k];
            if (offset + k < cap) {
                if (desc) compute_orb_descriptor(&kp, e->blurred[level], lw, bit_pattern_31, desc + (size_t)(offset + k) * 32);
                if (kps) {
                    if (level != 0) {
                        float scale = e->mvScaleFactor[level];
                        kp.x *= scale;
                        kp.y *= scale;
                    }
                    kps[offset + k] = kp;
                }
            } else
                status = -2;
        }
        offset += nk;
    }
    return status;
}
