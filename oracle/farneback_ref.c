/*
 * oracle/farneback_ref.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, single thread) of dense Farneback optical flow as the
 * reference obtains it from `cv2.calcOpticalFlowFarneback` (12 call sites, e.g.
 * /root/reference/optical_flow_seg.py:158,203,494; parameters
 * /root/reference/optical_flow_seg.py:73-81; the pinned dependency is
 * opencv-python==4.9.0, /root/reference/requirements.txt:1).
 *
 * The arithmetic lives in that third-party wheel (upstream file
 * modules/video/src/optflowgf.cpp plus GaussianBlur / resize / convertTo from
 * imgproc/core), which is NOT present in /root/reference nor importable in the build
 * container.  This file restates the *published* algorithm of the generic (non-IPP,
 * non-OpenCL, non-FMA) C++ path: same operation order, same float / double placement.
 *
 * PARITY UNPINNED AT BIT LEVEL: the reference holds no golden vectors, fixtures or tests
 * for this boundary and cv2 cannot be run here.  What does
 * tie it to real cv2 output is the one recorded result the reference holds: the figure in
 * the last cell of demo.ipynb (colour-coded flow of grasp 1.jpg -> 2.jpg, params A, drawn
 * at 247x438 px).  Rendered the same way, this restatement agrees with that panel to
 * 0.31 grey levels mean / 1 at the 99th percentile / correlation 0.9995, at the floor of the
 * comparison itself, and nearby parameter sets fail the same test (tests/test_demo_pin.py).
 * Beyond that it is pinned by analytic properties (tests/test_oracle_farneback.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off: no FMA contraction, matching
 * an SSE2/SSE3 baseline x86-64 OpenCV build).
 */
#include <math.h>
#include <float.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define NSOF_REF_OK 0
#define NSOF_REF_EINVAL (-1)
#define NSOF_REF_ENOMEM (-4)
#define NSOF_REF_EUNSUPPORTED (-5)

/* ---- small OpenCV primitives ------------------------------------------------------ */

/* cvRound on x86-64 = cvtsd2si = round-half-to-even in the default rounding mode. */
static int cv_round(double v) { return (int)lrint(v); }

/* cvFloor(float): int i = (int)value; return i - (i > value); */
static int cv_floor_f(float v)
{
    int i = (int)v;
    return i - (i > v);
}

/* borderInterpolate(p, len, BORDER_REFLECT_101) */
static int reflect101(int p, int len)
{
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    do {
        if (p < 0) p = -p;
        else p = 2 * len - 2 - p;
    } while ((unsigned)p >= (unsigned)len);
    return p;
}

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* ---- getGaussianKernel(n, sigma, CV_32F) -------------------------------------------
 * 4.x computes the taps in (soft)double: t_i = exp(-(i-(n-1)/2)^2 / (2 sigma^2)),
 * normalises by 1/sum in double and only then narrows to float.  sigma <= 0 selects
 * the fixed small tables ([.25,.5,.25] for n = 3). */
int nsof_ref_gaussian_kernel(int n, double sigma, float* out)
{
    static const double tab1[] = {1.0};
    static const double tab3[] = {0.25, 0.5, 0.25};
    static const double tab5[] = {0.0625, 0.25, 0.375, 0.25, 0.0625};
    static const double tab7[] = {0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125};
    static const double tab9[] = {4.0 / 256, 13.0 / 256, 30.0 / 256, 51.0 / 256, 60.0 / 256,
                                  51.0 / 256, 30.0 / 256, 13.0 / 256, 4.0 / 256};
    if (n <= 0 || (n & 1) == 0) return NSOF_REF_EINVAL;
    if (sigma <= 0) {
        const double* t = n == 1 ? tab1 : n == 3 ? tab3 : n == 5 ? tab5 : n == 7 ? tab7 : n == 9 ? tab9 : NULL;
        if (t) {
            for (int i = 0; i < n; i++) out[i] = (float)t[i];
            return NSOF_REF_OK;
        }
    }
    double sigmaX = sigma > 0 ? sigma : ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
    double scale2X = -0.125 / (sigmaX * sigmaX);
    int n2 = (n - 1) / 2;
    double* v = (double*)malloc(sizeof(double) * (size_t)(n2 + 1));
    if (!v) return NSOF_REF_ENOMEM;
    double sum = 0;
    for (int i = 0, x = 1 - n; i < n2; i++, x += 2) {
        double t = exp((double)(x * x) * scale2X);
        v[i] = t;
        sum += t;
    }
    sum *= 2.0;
    sum += 1.0;
    double mul1 = 1.0 / sum;
    for (int i = 0; i < n2; i++) {
        double t = v[i] * mul1;
        out[i] = (float)t;
        out[n - 1 - i] = (float)t;
    }
    out[n2] = (float)(1.0 * mul1);
    free(v);
    return NSOF_REF_OK;
}

/* ---- arithmetic variant of the pyramid stages ----------------------------------------------
 * The float Gaussian blur and the bilinear resamples below exist in two variants.  Variant 0 (default) rounds every
 * product and every sum, as the library's generic C++ loops do when they are compiled without contraction.  Variant 1
 * evaluates the same taps in the same order with ONE fused multiply-add per tap / blend (the leading product still
 * rounded): how an AVX2+FMA3 build of the library's vector loops (v_muladd in its separable-filter and resize kernels)
 * contracts them.  Which one a given cv2 wheel executes cannot be pinned in this image; DESIGN.md section 2 states how
 * far the flow moves between the two.  nsof_ref_set_pyr_fma selects the variant for subsequent calls (process-wide). */
static int g_pyr_fma = 0;
void nsof_ref_set_pyr_fma(int on) { g_pyr_fma = on ? 1 : 0; }
int nsof_ref_get_pyr_fma(void) { return g_pyr_fma; }
static inline float madd(float a, float b, float c) { return g_pyr_fma ? fmaf(a, b, c) : a * b + c; }

/* ---- GaussianBlur on CV_32FC1 via sepFilter2D, BORDER_REFLECT_101 -------------------
 * Row pass first (float accumulation), then column pass (float accumulation).
 *   row, ksize <= 5 (SymmRowSmallFilter):  S0*k0 + (S-1 + S1)*k1 [+ (S-2 + S2)*k2]
 *   row, ksize  > 5 (RowFilter):           left-to-right single taps, starting at tap 0
 *   col, ksize == 3 (SymmColumnSmallFilter): (Sm + Sp)*k1 + S0*k0
 *   col, ksize  > 3 (SymmColumnFilter):    S0*k0, then += k_j*(S+j + S-j), j = 1..r   */
static int gaussian_blur_f32(const float* src, int w, int h, float* dst, int ksize, double sigma)
{
    if (ksize == 1) {
        memcpy(dst, src, sizeof(float) * (size_t)w * h);
        return NSOF_REF_OK;
    }
    float kbuf[64];
    if (ksize > 63) return NSOF_REF_EINVAL;
    int rc = nsof_ref_gaussian_kernel(ksize, sigma, kbuf);
    if (rc) return rc;
    const int r = ksize / 2;
    const float* kc = kbuf + r; /* centre */
    float* tmp = (float*)malloc(sizeof(float) * (size_t)w * h);
    int* xi = (int*)malloc(sizeof(int) * (size_t)(w + 2 * r));
    if (!tmp || !xi) { free(tmp); free(xi); return NSOF_REF_ENOMEM; }
    for (int x = -r; x < w + r; x++) xi[x + r] = reflect101(x, w);

    for (int y = 0; y < h; y++) {
        const float* S = src + (size_t)y * w;
        float* D = tmp + (size_t)y * w;
        for (int x = 0; x < w; x++) {
            const int* ix = xi + x + r; /* ix[j] = reflected x+j */
            float s;
            if (ksize == 3) {
                s = madd(S[ix[-1]] + S[ix[1]], kc[1], S[ix[0]] * kc[0]);
            } else if (ksize == 5) {
                s = madd(S[ix[-2]] + S[ix[2]], kc[2], madd(S[ix[-1]] + S[ix[1]], kc[1], S[ix[0]] * kc[0]));
            } else {
                s = kbuf[0] * S[ix[-r]];
                for (int k = 1; k < ksize; k++) s = madd(kbuf[k], S[ix[k - r]], s);
            }
            D[x] = s;
        }
    }
    for (int y = 0; y < h; y++) {
        float* D = dst + (size_t)y * w;
        const float* S0 = tmp + (size_t)y * w;
        if (ksize == 3) {
            const float* Sm = tmp + (size_t)reflect101(y - 1, h) * w;
            const float* Sp = tmp + (size_t)reflect101(y + 1, h) * w;
            for (int x = 0; x < w; x++) D[x] = madd(Sm[x] + Sp[x], kc[1], S0[x] * kc[0]);
        } else {
            for (int x = 0; x < w; x++) D[x] = kc[0] * S0[x];
            for (int k = 1; k <= r; k++) {
                const float* Sp = tmp + (size_t)reflect101(y + k, h) * w;
                const float* Sm = tmp + (size_t)reflect101(y - k, h) * w;
                for (int x = 0; x < w; x++) D[x] = madd(kc[k], Sp[x] + Sm[x], D[x]);
            }
        }
    }
    free(tmp);
    free(xi);
    return NSOF_REF_OK;
}

/* ---- resize(INTER_LINEAR) for CV_32FC(cn), generic path -------------------------------
 * scale = 1/((double)dst/src); f = (float)((d+0.5)*scale - 0.5); s = floor(f); f -= s.
 * Horizontal: s < 0 -> (s=0,f=0); s >= sw-1 -> (s=sw-1,f=0).  Vertical: weights are NOT
 * zeroed, the two row indices are clamped instead.  Horizontal pass first, float math,
 * D = S[s]*(1-f) + S[s+1]*f.  Same size -> plain copy. */
int nsof_ref_resize_linear(const float* src, int sw, int sh, int cn, float* dst, int dw, int dh)
{
    if (sw <= 0 || sh <= 0 || dw <= 0 || dh <= 0 || cn <= 0) return NSOF_REF_EINVAL;
    if (sw == dw && sh == dh) {
        memcpy(dst, src, sizeof(float) * (size_t)sw * sh * cn);
        return NSOF_REF_OK;
    }
    double inv_sx = (double)dw / sw, inv_sy = (double)dh / sh;
    double scale_x = 1. / inv_sx, scale_y = 1. / inv_sy;
    int* xofs = (int*)malloc(sizeof(int) * (size_t)dw);
    float* xa = (float*)malloc(sizeof(float) * (size_t)dw);
    float* hbuf = (float*)malloc(sizeof(float) * (size_t)dw * cn * sh); /* horizontal pass of every source row */
    if (!xofs || !xa || !hbuf) { free(xofs); free(xa); free(hbuf); return NSOF_REF_ENOMEM; }
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor_f(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        xa[dx] = fx;
    }
    for (int y = 0; y < sh; y++) {
        const float* S = src + (size_t)y * sw * cn;
        float* D = hbuf + (size_t)y * dw * cn;
        for (int dx = 0; dx < dw; dx++) {
            int sx = xofs[dx];
            float a1 = xa[dx], a0 = 1.f - a1;
            for (int c = 0; c < cn; c++) {
                if (sx + 1 < sw) D[dx * cn + c] = madd(S[sx * cn + c], a0, S[(sx + 1) * cn + c] * a1);
                else D[dx * cn + c] = S[sx * cn + c] * 1.f; /* dx >= xmax branch */
            }
        }
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor_f(fy);
        fy -= sy;
        float b0 = 1.f - fy, b1 = fy;
        const float* S0 = hbuf + (size_t)clampi(sy, 0, sh - 1) * dw * cn;
        const float* S1 = hbuf + (size_t)clampi(sy + 1, 0, sh - 1) * dw * cn;
        float* D = dst + (size_t)dy * dw * cn;
        for (int i = 0; i < dw * cn; i++) D[i] = madd(S0[i], b0, S1[i] * b1);
    }
    free(xofs); free(xa); free(hbuf);
    return NSOF_REF_OK;
}

/* ---- level geometry (driver loop of FarnebackOpticalFlowImpl::calc) -------------------- */
int nsof_ref_effective_levels(int w, int h, double pyr_scale, int levels)
{
    int k;
    double scale = 1;
    for (k = 0; k < levels; k++) {
        scale *= pyr_scale;
        if (w * scale < 32 || h * scale < 32) break;
    }
    return k;
}

/* out: w_k, h_k, smooth_sz, sigma for level k */
void nsof_ref_level_geometry(int w, int h, double pyr_scale, int k, int* wk, int* hk, int* ksize, double* sigma)
{
    double scale = 1;
    for (int i = 0; i < k; i++) scale *= pyr_scale;
    double s = (1. / scale - 1) * 0.5;
    int sz = cv_round(s * 5) | 1;
    if (sz < 3) sz = 3;
    *wk = cv_round(w * scale);
    *hk = cv_round(h * scale);
    *ksize = sz;
    *sigma = s;
}

/* convertTo(CV_32F) -> GaussianBlur -> resize for one level; out is hk x wk float. */
int nsof_ref_pyr_level(const uint8_t* img, ptrdiff_t stride, int w, int h, double pyr_scale, int k, float* out)
{
    int wk, hk, ksize; double sigma;
    nsof_ref_level_geometry(w, h, pyr_scale, k, &wk, &hk, &ksize, &sigma);
    float* f = (float*)malloc(sizeof(float) * (size_t)w * h);
    float* b = (float*)malloc(sizeof(float) * (size_t)w * h);
    if (!f || !b) { free(f); free(b); return NSOF_REF_ENOMEM; }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) f[(size_t)y * w + x] = (float)img[(ptrdiff_t)y * stride + x];
    int rc = gaussian_blur_f32(f, w, h, b, ksize, sigma);
    if (!rc) rc = nsof_ref_resize_linear(b, w, h, 1, out, wk, hk);
    free(f); free(b);
    return rc;
}

/* ---- FarnebackPrepareGaussian --------------------------------------------------------
 * n is a RADIUS (taps -n..n).  g/xg/xxg are float; G is 6x6 double; invG by Cholesky. */
static int cholesky_inverse6(double A[6][6], double inv[6][6])
{
    double L[6][6];
    memset(L, 0, sizeof(L));
    for (int i = 0; i < 6; i++) {
        for (int j = 0; j <= i; j++) {
            double s = A[i][j];
            for (int k = 0; k < j; k++) s -= L[i][k] * L[j][k];
            if (i == j) {
                if (s <= 0) return NSOF_REF_EINVAL;
                L[i][i] = sqrt(s);
            } else {
                L[i][j] = s / L[j][j];
            }
        }
    }
    for (int c = 0; c < 6; c++) {
        double y[6], x[6];
        for (int i = 0; i < 6; i++) {
            double s = (i == c) ? 1.0 : 0.0;
            for (int k = 0; k < i; k++) s -= L[i][k] * y[k];
            y[i] = s / L[i][i];
        }
        for (int i = 5; i >= 0; i--) {
            double s = y[i];
            for (int k = i + 1; k < 6; k++) s -= L[k][i] * x[k];
            x[i] = s / L[i][i];
        }
        for (int i = 0; i < 6; i++) inv[i][c] = x[i];
    }
    return NSOF_REF_OK;
}

/* g, xg, xxg: arrays of 2n+1 floats indexed [x+n]; ig: {ig11, ig03, ig33, ig55} */
int nsof_ref_poly_prepare(int n, double sigma, float* g, float* xg, float* xxg, double* ig)
{
    if (n < 1) return NSOF_REF_EINVAL;
    if (sigma < FLT_EPSILON) sigma = n * 0.3;
    double s = 0.;
    for (int x = -n; x <= n; x++) {
        g[x + n] = (float)exp(-x * x / (2 * sigma * sigma));
        s += g[x + n];
    }
    s = 1. / s;
    for (int x = -n; x <= n; x++) {
        g[x + n] = (float)(g[x + n] * s);
        xg[x + n] = (float)(x * g[x + n]);
        xxg[x + n] = (float)(x * x * g[x + n]);
    }
    double G[6][6], invG[6][6];
    memset(G, 0, sizeof(G));
    for (int y = -n; y <= n; y++)
        for (int x = -n; x <= n; x++) {
            G[0][0] += g[y + n] * g[x + n];
            G[1][1] += g[y + n] * g[x + n] * x * x;
            G[3][3] += g[y + n] * g[x + n] * x * x * x * x;
            G[5][5] += g[y + n] * g[x + n] * x * x * y * y;
        }
    G[2][2] = G[0][3] = G[0][4] = G[3][0] = G[4][0] = G[1][1];
    G[4][4] = G[3][3];
    G[3][4] = G[4][3] = G[5][5];
    int rc = cholesky_inverse6(G, invG);
    if (rc) return rc;
    ig[0] = invG[1][1];
    ig[1] = invG[0][3];
    ig[2] = invG[3][3];
    ig[3] = invG[5][5];
    return NSOF_REF_OK;
}

/* ---- FarnebackPolyExp: src h x w float -> dst h x w x 5 float (interleaved) -------------
 * Vertical pass in float with replicate rows; horizontal pass accumulates in double,
 * but note which products are formed in float (b2,b3,b5,b6) and which in double (b1,b4). */
int nsof_ref_polyexp(const float* src, int width, int height, int n, double sigma, float* dst)
{
    if (n < 1 || width < 1 || height < 1) return NSOF_REF_EINVAL;
    float* kbuf = (float*)malloc(sizeof(float) * (size_t)(n * 6 + 3));
    float* _row = (float*)malloc(sizeof(float) * (size_t)(width + n * 2) * 3);
    if (!kbuf || !_row) { free(kbuf); free(_row); return NSOF_REF_ENOMEM; }
    float* g = kbuf + n;
    float* xg = g + n * 2 + 1;
    float* xxg = xg + n * 2 + 1;
    float* row = _row + n * 3;
    double ig[4];
    int rc = nsof_ref_poly_prepare(n, sigma, kbuf, kbuf + 2 * n + 1, kbuf + 4 * n + 2, ig);
    if (rc) { free(kbuf); free(_row); return rc; }
    const double ig11 = ig[0], ig03 = ig[1], ig33 = ig[2], ig55 = ig[3];

    for (int y = 0; y < height; y++) {
        float g0 = g[0], g1, g2;
        const float* srow0 = src + (size_t)y * width;
        const float* srow1;
        float* drow = dst + (size_t)y * width * 5;

        for (int x = 0; x < width; x++) {
            row[x * 3] = srow0[x] * g0;
            row[x * 3 + 1] = row[x * 3 + 2] = 0.f;
        }
        for (int k = 1; k <= n; k++) {
            g0 = g[k]; g1 = xg[k]; g2 = xxg[k];
            srow0 = src + (size_t)(y - k > 0 ? y - k : 0) * width;
            srow1 = src + (size_t)(y + k < height - 1 ? y + k : height - 1) * width;
            for (int x = 0; x < width; x++) {
                float p = srow0[x] + srow1[x];
                float t0 = row[x * 3] + g0 * p;
                float t1 = row[x * 3 + 2] + g2 * p;
                row[x * 3] = t0;
                row[x * 3 + 2] = t1;
                p = srow1[x] - srow0[x];
                t0 = row[x * 3 + 1] + g1 * p;
                row[x * 3 + 1] = t0;
            }
        }
        for (int x = 0; x < n * 3; x++) {
            row[-1 - x] = row[2 - x];
            row[width * 3 + x] = row[width * 3 + x - 3];
        }
        for (int x = 0; x < width; x++) {
            g0 = g[0];
            double b1 = row[x * 3] * g0, b2 = 0, b3 = row[x * 3 + 1] * g0, b4 = 0, b5 = row[x * 3 + 2] * g0, b6 = 0;
            for (int k = 1; k <= n; k++) {
                double tg = row[(x + k) * 3] + row[(x - k) * 3];
                g0 = g[k];
                b1 += tg * g0;
                b4 += tg * xxg[k];
                b2 += (row[(x + k) * 3] - row[(x - k) * 3]) * xg[k];
                b3 += (row[(x + k) * 3 + 1] + row[(x - k) * 3 + 1]) * g0;
                b6 += (row[(x + k) * 3 + 1] - row[(x - k) * 3 + 1]) * xg[k];
                b5 += (row[(x + k) * 3 + 2] + row[(x - k) * 3 + 2]) * g0;
            }
            drow[x * 5 + 1] = (float)(b2 * ig11);
            drow[x * 5] = (float)(b3 * ig11);
            drow[x * 5 + 3] = (float)(b1 * ig03 + b4 * ig33);
            drow[x * 5 + 2] = (float)(b1 * ig03 + b5 * ig33);
            drow[x * 5 + 4] = (float)(b6 * ig55);
        }
    }
    free(kbuf); free(_row);
    return NSOF_REF_OK;
}

/* ---- FarnebackUpdateMatrices: rows [y0,y1) of M (h x w x 5 float) ------------------------- */
int nsof_ref_update_matrices(const float* R0a, const float* R1a, const float* flowa, float* Ma,
                             int width, int height, int _y0, int _y1)
{
    enum { BORDER = 5 };
    static const float border[BORDER] = {0.14f, 0.14f, 0.4472f, 0.4472f, 0.4472f};
    const size_t step1 = (size_t)width * 5;
    for (int y = _y0; y < _y1; y++) {
        const float* flow = flowa + (size_t)y * width * 2;
        const float* R0 = R0a + (size_t)y * width * 5;
        float* M = Ma + (size_t)y * width * 5;
        for (int x = 0; x < width; x++) {
            float dx = flow[x * 2], dy = flow[x * 2 + 1];
            float fx = x + dx, fy = y + dy;
            int x1 = cv_floor_f(fx), y1 = cv_floor_f(fy);
            float r2, r3, r4, r5, r6;
            fx -= x1; fy -= y1;
            if ((unsigned)x1 < (unsigned)(width - 1) && (unsigned)y1 < (unsigned)(height - 1)) {
                const float* ptr = R1a + (size_t)y1 * step1 + (size_t)x1 * 5;
                float a00 = (1.f - fx) * (1.f - fy), a01 = fx * (1.f - fy), a10 = (1.f - fx) * fy, a11 = fx * fy;
                r2 = a00 * ptr[0] + a01 * ptr[5] + a10 * ptr[step1] + a11 * ptr[step1 + 5];
                r3 = a00 * ptr[1] + a01 * ptr[6] + a10 * ptr[step1 + 1] + a11 * ptr[step1 + 6];
                r4 = a00 * ptr[2] + a01 * ptr[7] + a10 * ptr[step1 + 2] + a11 * ptr[step1 + 7];
                r5 = a00 * ptr[3] + a01 * ptr[8] + a10 * ptr[step1 + 3] + a11 * ptr[step1 + 8];
                r6 = a00 * ptr[4] + a01 * ptr[9] + a10 * ptr[step1 + 4] + a11 * ptr[step1 + 9];
                r4 = (R0[x * 5 + 2] + r4) * 0.5f;
                r5 = (R0[x * 5 + 3] + r5) * 0.5f;
                r6 = (R0[x * 5 + 4] + r6) * 0.25f;
            } else {
                r2 = r3 = 0.f;
                r4 = R0[x * 5 + 2];
                r5 = R0[x * 5 + 3];
                r6 = R0[x * 5 + 4] * 0.5f;
            }
            r2 = (R0[x * 5] - r2) * 0.5f;
            r3 = (R0[x * 5 + 1] - r3) * 0.5f;
            r2 += r4 * dy + r6 * dx;
            r3 += r6 * dy + r5 * dx;
            if ((unsigned)(x - BORDER) >= (unsigned)(width - BORDER * 2) ||
                (unsigned)(y - BORDER) >= (unsigned)(height - BORDER * 2)) {
                float scale = (x < BORDER ? border[x] : 1.f) * (x >= width - BORDER ? border[width - x - 1] : 1.f) *
                              (y < BORDER ? border[y] : 1.f) * (y >= height - BORDER ? border[height - y - 1] : 1.f);
                r2 *= scale; r3 *= scale; r4 *= scale; r5 *= scale; r6 *= scale;
            }
            M[x * 5] = r4 * r4 + r6 * r6;
            M[x * 5 + 1] = (r4 + r5) * r6;
            M[x * 5 + 2] = r5 * r5 + r6 * r6;
            M[x * 5 + 3] = r4 * r2 + r6 * r3;
            M[x * 5 + 4] = r6 * r2 + r5 * r3;
        }
    }
    return NSOF_REF_OK;
}

/* ---- FarnebackUpdateFlow_Blur -----------------------------------------------------------
 * Box (2m+1)^2, m = block_size/2, replicate borders, normalised by 1/block_size^2.
 * Column sums: double running sums of FLOAT differences (srow1[x]-srow0[x] is rounded to
 * float before it is added).  Row sums: double running sums of double differences.
 * The lagged-stripe matrix update is kept exactly as upstream. */
int nsof_ref_update_flow_blur(const float* R0, const float* R1, float* flowa, float* Ma,
                              int width, int height, int block_size, int update_matrices)
{
    int m = block_size / 2;
    int y0 = 0, y1;
    int min_update_stripe = (1 << 10) / width > block_size ? (1 << 10) / width : block_size;
    double scale = 1. / (block_size * block_size);
    double* _vsum = (double*)malloc(sizeof(double) * (size_t)(width + m * 2 + 2) * 5);
    if (!_vsum) return NSOF_REF_ENOMEM;
    double* vsum = _vsum + (m + 1) * 5;

    const float* srow0 = Ma;
    for (int x = 0; x < width * 5; x++) vsum[x] = srow0[x] * (m + 2);
    for (int y = 1; y < m; y++) {
        srow0 = Ma + (size_t)(y < height - 1 ? y : height - 1) * width * 5;
        for (int x = 0; x < width * 5; x++) vsum[x] += srow0[x];
    }
    for (int y = 0; y < height; y++) {
        double g11, g12, g22, h1, h2;
        float* flow = flowa + (size_t)y * width * 2;
        srow0 = Ma + (size_t)(y - m - 1 > 0 ? y - m - 1 : 0) * width * 5;
        const float* srow1 = Ma + (size_t)(y + m < height - 1 ? y + m : height - 1) * width * 5;
        for (int x = 0; x < width * 5; x++) vsum[x] += srow1[x] - srow0[x];
        for (int x = 0; x < (m + 1) * 5; x++) {
            vsum[-1 - x] = vsum[4 - x];
            vsum[width * 5 + x] = vsum[width * 5 + x - 5];
        }
        g11 = vsum[0] * (m + 2);
        g12 = vsum[1] * (m + 2);
        g22 = vsum[2] * (m + 2);
        h1 = vsum[3] * (m + 2);
        h2 = vsum[4] * (m + 2);
        for (int x = 1; x < m; x++) {
            g11 += vsum[x * 5];
            g12 += vsum[x * 5 + 1];
            g22 += vsum[x * 5 + 2];
            h1 += vsum[x * 5 + 3];
            h2 += vsum[x * 5 + 4];
        }
        for (int x = 0; x < width; x++) {
            g11 += vsum[(x + m) * 5] - vsum[(x - m) * 5 - 5];
            g12 += vsum[(x + m) * 5 + 1] - vsum[(x - m) * 5 - 4];
            g22 += vsum[(x + m) * 5 + 2] - vsum[(x - m) * 5 - 3];
            h1 += vsum[(x + m) * 5 + 3] - vsum[(x - m) * 5 - 2];
            h2 += vsum[(x + m) * 5 + 4] - vsum[(x - m) * 5 - 1];
            double g11_ = g11 * scale, g12_ = g12 * scale, g22_ = g22 * scale;
            double h1_ = h1 * scale, h2_ = h2 * scale;
            double idet = 1. / (g11_ * g22_ - g12_ * g12_ + 1e-3);
            flow[x * 2] = (float)((g11_ * h2_ - g12_ * h1_) * idet);
            flow[x * 2 + 1] = (float)((g22_ * h1_ - g12_ * h2_) * idet);
        }
        y1 = y == height - 1 ? height : y - block_size;
        if (update_matrices && (y1 == height || y1 >= y0 + min_update_stripe)) {
            nsof_ref_update_matrices(R0, R1, flowa, Ma, width, height, y0, y1);
            y0 = y1;
        }
    }
    free(_vsum);
    return NSOF_REF_OK;
}

/* ---- FarnebackOpticalFlowImpl::calc (flags == 0 path) ------------------------------------ */
int nsof_ref_farneback_u8(const uint8_t* prev, ptrdiff_t prev_stride, const uint8_t* next, ptrdiff_t next_stride,
                          int width, int height, float* flow_out, ptrdiff_t flow_stride_bytes,
                          double pyr_scale, int levels, int winsize, int iterations, int poly_n,
                          double poly_sigma, int flags)
{
    if (!prev || !next || !flow_out || width < 1 || height < 1) return NSOF_REF_EINVAL;
    if (!(pyr_scale < 1) || !(pyr_scale > 0) || levels < 0 || winsize < 1 || iterations < 0 || poly_n < 1)
        return NSOF_REF_EINVAL;
    if (flags != 0) return NSOF_REF_EUNSUPPORTED; /* USE_INITIAL_FLOW / GAUSSIAN: never used by the reference */

    const uint8_t* img[2] = {prev, next};
    const ptrdiff_t stride[2] = {prev_stride, next_stride};
    levels = nsof_ref_effective_levels(width, height, pyr_scale, levels);
    const size_t n0 = (size_t)width * height;
    float* prevFlow = NULL; int pw = 0, ph = 0;
    float* I = (float*)malloc(sizeof(float) * n0);
    float* R[2] = {(float*)malloc(sizeof(float) * n0 * 5), (float*)malloc(sizeof(float) * n0 * 5)};
    float* M = (float*)malloc(sizeof(float) * n0 * 5);
    int rc = (I && R[0] && R[1] && M) ? NSOF_REF_OK : NSOF_REF_ENOMEM;

    for (int k = levels; k >= 0 && !rc; k--) {
        int w, h, ksize; double sigma;
        nsof_ref_level_geometry(width, height, pyr_scale, k, &w, &h, &ksize, &sigma);
        float* flow = (float*)malloc(sizeof(float) * (size_t)w * h * 2);
        if (!flow) { rc = NSOF_REF_ENOMEM; break; }
        if (!prevFlow) {
            memset(flow, 0, sizeof(float) * (size_t)w * h * 2);
        } else {
            rc = nsof_ref_resize_linear(prevFlow, pw, ph, 2, flow, w, h);
            /* flow *= 1./pyr_scale -> convertTo(-1, alpha): float(alpha), float multiply */
            float a = (float)(1. / pyr_scale);
            for (size_t i = 0; i < (size_t)w * h * 2; i++) flow[i] = flow[i] * a;
        }
        for (int i = 0; i < 2 && !rc; i++) {
            rc = nsof_ref_pyr_level(img[i], stride[i], width, height, pyr_scale, k, I);
            if (!rc) rc = nsof_ref_polyexp(I, w, h, poly_n, poly_sigma, R[i]);
        }
        if (!rc) rc = nsof_ref_update_matrices(R[0], R[1], flow, M, w, h, 0, h);
        for (int i = 0; i < iterations && !rc; i++)
            rc = nsof_ref_update_flow_blur(R[0], R[1], flow, M, w, h, winsize, i < iterations - 1);
        free(prevFlow);
        prevFlow = flow; pw = w; ph = h;
    }
    if (!rc) {
        for (int y = 0; y < height; y++)
            memcpy((char*)flow_out + (ptrdiff_t)y * flow_stride_bytes, prevFlow + (size_t)y * width * 2,
                   sizeof(float) * (size_t)width * 2);
    }
    free(prevFlow); free(I); free(R[0]); free(R[1]); free(M);
    return rc;
}

/* Many independent pairs of one shape: the CPU-baseline legs of bench.py time this (one thread, and -- in the
 * OpenMP build `make perf` -- all host cores, one pair per thread at a time: pairs are independent, exactly how
 * the GPU path shards them).  prev/next: [n][height][stride]; flow: [n][height][width][2]. */
int nsof_ref_farneback_u8_many(int n, const uint8_t* prev, const uint8_t* next, ptrdiff_t stride, ptrdiff_t pair_stride,
                               int width, int height, float* flow, double pyr_scale, int levels, int winsize,
                               int iterations, int poly_n, double poly_sigma, int flags, int n_threads)
{
    int rc_all = NSOF_REF_OK;
    (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads > 0 ? n_threads : 1)
#endif
    for (int i = 0; i < n; i++) {
        int rc = nsof_ref_farneback_u8(prev + (ptrdiff_t)i * pair_stride, stride, next + (ptrdiff_t)i * pair_stride, stride,
                                       width, height, flow + (size_t)i * width * height * 2, (ptrdiff_t)width * 8,
                                       pyr_scale, levels, winsize, iterations, poly_n, poly_sigma, flags);
        if (rc) {
#ifdef _OPENMP
#pragma omp critical
#endif
            rc_all = rc;
        }
    }
    return rc_all;
}

/* 1 when this library was built with OpenMP (the all-cores leg needs the `perf` build). */
int nsof_ref_has_openmp(void)
{
#ifdef _OPENMP
    return 1;
#else
    return 0;
#endif
}
