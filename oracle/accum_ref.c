/*
 * oracle/accum_ref.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, single thread) of the reference's event-driven synaptic
 * accumulator, /root/reference/eventsim/event_mem_sim.py:
 *   PARAMS / DT / THETA_EVENTS / REFRACTORY_US   :20-34
 *   update_state                                 :40-57
 *   resistance_exp                               :60-63
 *   slice_indices                                :78-83
 *   simulate (scheme 1, scheme 2 split/magnitude):164-286
 *
 * PINNED: tests/golden/accum_*.npz were produced by importing and running that file
 * in the build container (tests/golden/gen_accum_golden.py); tests/test_oracle_accum.py
 * checks this restatement against them.  float32 arithmetic follows NumPy's casting of
 * python-float constants to float32; powf / expf come from libm where NumPy may use its
 * SIMD (SVML / npyv) kernels, hence the ulp-level tolerance stated in the tests.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* event_mem_sim.py:20-27 (python floats, narrowed to float32 by NumPy at use) */
static const float VOFF = (float)-0.2, VON = (float)0.1;
static const float KOFF = (float)51.03, KON = (float)-2.91;
static const float SON = (float)0.2, SOFF = (float)0.8;
static const float BON = (float)-5.12, BOFF = (float)3.10;
static const float DT = (float)5e-4;        /* :30 */
static const double RON = 163305.0, ROFF = 2104377.0;
#define THETA_EVENTS 1   /* :33 */
#define REFRACTORY_US 800 /* :34 */

/* update_state (:40-57); alphaoff = alphaon = 1 so "** alpha" is the identity. */
static inline float update_one(float w, float V)
{
    float dwdt = 0.f;
    if (V < VOFF) {
        float a = V / VOFF - 1.f;
        float b = powf(1.f - w * SOFF, BOFF);
        dwdt = KOFF * a * b;
    } else if (V > VON) {
        float a = V / VON - 1.f;
        float b = powf(1.f - w * SON, BON);
        dwdt = KON * a * b;
    }
    float wn = w + dwdt * DT;
    return wn < 0.f ? 0.f : (wn > 1.f ? 1.f : wn);
}

/* Threads of the dense passes below: 1 unless the OpenMP build (`make perf`, bench.py's all-cores CPU leg) raises it. */
static int g_threads = 1;
void nsof_ref_set_threads(int n) { g_threads = n > 1 ? n : 1; }

void nsof_ref_accum_update_state(const float* w, const float* V, float* out, size_t n)
{
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1)
#endif
    for (size_t i = 0; i < n; i++) out[i] = update_one(w[i], V[i]);
}

/* One time slice of scheme 1 as the reference runs it (event_mem_sim.py:208-220): V = silent_v everywhere, active_v at
 * the slice's event pixels, then the dense update -- the unit bench.py's CPU leg times (slices/s). */
void nsof_ref_accum_slice_v1(float* w, float* V, size_t npx, const int16_t* x, const int16_t* y, int64_t n_ev, int W,
                             float active_v, float silent_v)
{
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1)
#endif
    for (size_t i = 0; i < npx; i++) V[i] = silent_v;
    for (int64_t e = 0; e < n_ev; e++) V[(size_t)y[e] * W + x[e]] = active_v;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1)
#endif
    for (size_t i = 0; i < npx; i++) w[i] = update_one(w[i], V[i]);
}

/* resistance_exp (:60-63): lam = log(Roff/Ron) (float64 scalar); exp argument and exp in
 * float32 (NumPy 1.x: a float64 *scalar* does not upcast a float32 array); Ron / exp(..)
 * is a float64 division (python int 163305 -> uint32 -> promote(uint32,float32)=float64),
 * narrowed to float32 by the caller (:292). */
void nsof_ref_accum_resistance(const float* w, float* out, size_t n)
{
    const float neg_lam = (float)(-log(ROFF / RON));
    for (size_t i = 0; i < n; i++) {
        float e = expf(neg_lam * (1.0f - w[i]));
        out[i] = (float)(RON / (double)e); /* python-int Ron promotes the division to float64 (NumPy 1.x) */
    }
}

/* slice_indices (:78-83): bounds = arange(t[0], t[-1]+slice_us, slice_us);
 * idx = searchsorted(t, bounds, 'left').  Returns len(idx); fills idx if cap allows. */
int64_t nsof_ref_accum_slice_bounds(const int64_t* t, int64_t n, int64_t slice_us, int64_t* idx, int64_t cap)
{
    if (n <= 0 || slice_us <= 0) return 0;
    int64_t start = t[0], stop = t[n - 1] + slice_us;
    int64_t nb = (stop - start + slice_us - 1) / slice_us;
    if (nb < 0) nb = 0;
    if (idx) {
        int64_t pos = 0;
        for (int64_t i = 0; i < nb && i < cap; i++) {
            int64_t b = start + i * slice_us;
            while (pos < n && t[pos] < b) pos++;
            idx[i] = pos;
        }
    }
    return nb;
}

/* simulate (:164-286).  split != 0 only for version 2 / polarity 'split'.
 * Snapshots every max(1, nslices/100) slices (:181-183, :222, :277). */
int nsof_ref_accum_simulate(const int16_t* x, const int16_t* y, const int8_t* pol, const int64_t* t, int64_t n,
                            int H, int W, int version, int split, int64_t slice_us, float active_v, float silent_v,
                            float* w_a, float* res_a, float* w_b, float* res_b, int64_t nsnap_cap)
{
    if (version != 1 && version != 2) return -1;
    const size_t npx = (size_t)H * W;
    int64_t nb = nsof_ref_accum_slice_bounds(t, n, slice_us, NULL, 0);
    int64_t* idx = (int64_t*)malloc(sizeof(int64_t) * (size_t)(nb > 0 ? nb : 1));
    float* Va = (float*)malloc(sizeof(float) * npx);
    float* Vb = (float*)malloc(sizeof(float) * npx);
    int64_t* ok_a = (int64_t*)calloc(npx, sizeof(int64_t));
    int64_t* ok_b = (int64_t*)calloc(npx, sizeof(int64_t));
    uint8_t* elig = (uint8_t*)malloc((size_t)(n > 0 ? n : 1));
    if (!idx || !Va || !Vb || !ok_a || !ok_b || !elig) {
        free(idx); free(Va); free(Vb); free(ok_a); free(ok_b); free(elig);
        return -4;
    }
    nsof_ref_accum_slice_bounds(t, n, slice_us, idx, nb);
    const int64_t nslices = nb > 0 ? nb - 1 : 0;
    const int64_t every = nslices / 100 > 1 ? nslices / 100 : 1;
    for (size_t i = 0; i < npx; i++) w_a[i] = 0.5f; /* wini */
    if (split) for (size_t i = 0; i < npx; i++) w_b[i] = 0.5f;
    int64_t snap = 0;
    int rc = 0;

    for (int64_t s = 0; s < nslices; s++) {
        const int64_t lo = idx[s], hi = idx[s + 1];
        for (size_t i = 0; i < npx; i++) Va[i] = silent_v;
        if (version == 1) {
            /* bincount >= THETA_EVENTS(1)  <=>  at least one event on the pixel */
            for (int64_t e = lo; e < hi; e++) Va[(size_t)y[e] * W + x[e]] = active_v;
        } else {
            if (split) for (size_t i = 0; i < npx; i++) Vb[i] = silent_v;
            if (hi > lo) {
                const int64_t t_first = t[lo], t_next = t[hi - 1] + REFRACTORY_US;
                /* eligibility is evaluated for all events before any next_ok update */
                for (int64_t e = lo; e < hi; e++) {
                    size_t q = (size_t)y[e] * W + x[e];
                    if (split) {
                        if (pol[e] == 1) elig[e] = ok_a[q] <= t_first;
                        else if (pol[e] == 0) elig[e] = ok_b[q] <= t_first;
                        else elig[e] = 0;
                    } else {
                        elig[e] = ok_a[q] <= t_first;
                    }
                }
                for (int64_t e = lo; e < hi; e++) {
                    if (!elig[e]) continue;
                    size_t q = (size_t)y[e] * W + x[e];
                    if (split && pol[e] == 0) { Vb[q] = silent_v + active_v; ok_b[q] = t_next; }
                    else                      { Va[q] = silent_v + active_v; ok_a[q] = t_next; }
                }
            }
        }
        for (size_t i = 0; i < npx; i++) w_a[i] = update_one(w_a[i], Va[i]);
        if (split) for (size_t i = 0; i < npx; i++) w_b[i] = update_one(w_b[i], Vb[i]);
        if (s % every == 0) {
            if (snap >= nsnap_cap) { rc = -1; break; }
            nsof_ref_accum_resistance(w_a, res_a + (size_t)snap * npx, npx);
            if (split) nsof_ref_accum_resistance(w_b, res_b + (size_t)snap * npx, npx);
            snap++;
        }
    }
    free(idx); free(Va); free(Vb); free(ok_a); free(ok_b); free(elig);
    return rc;
}

/* ---- frame-driven variant: /root/reference/simulation/simulationcode_v4_transistor_uav.m -------------------
 *   calculate_difference_matrix + func1/2/3  :146-171     modulatefunc :332-347
 *   update_state (scalar, float64)           :173-184     simulate_memristor_array :187-227
 *   calculate_resistances_exp                :233-236
 * PARITY UNPINNED: there is no MATLAB/Octave in the build container and the reference ships no outputs of this
 * script; the restatement follows the source text.  `imgs` are the already compressed frames (doubles in [0,1],
 * the output of compress_image :111-121 -- imresize(...,'lanczos3') itself is not restated).
 * res holds n_frames snapshots: the initial array, then one after every frame pair. */
static double frame_update(double w, double V, double dt)
{
    double dwdt = 0.0;
    if (V < -0.2) dwdt = 51.03 * (V / -0.2 - 1) * pow(1 - w * 0.8, 3.10);
    else if (V > 0.1) dwdt = -2.91 * (V / 0.1 - 1) * pow(1 - w * 0.2, -5.12);
    double nw = w + dwdt * dt;
    return nw < 0 ? 0 : (nw > 1 ? 1 : nw);
}

double nsof_ref_frame_drive(double a, double b, double th1, double th2)
{
    double d = fabs(a * 256 - b * 256), V;
    /* three masked assignments in the source, the last one (diff <= th1 -> func1) wins where masks overlap
     * (the vehicle script has th1 = 2 > th2 = 1.5) */
    if (d <= th1) V = (d - 5.5) * 0.6;          /* func1 */
    else if (d <= th2) V = (d + 4) * 0.75;      /* func2 */
    else V = (d + 4) * 0.75;                    /* func3 */
    /* modulatefunc: undefined for V == 0 in the source (v_mod unset); 0 is returned here */
    if (V > 0) return -(0.3 * V + 0);
    if (V < 0) return -(3 * V + -3);
    return 0.0;
}

int nsof_ref_accum_frames(const double* imgs, int n_frames, int H, int W, double dt, int n_sub, double th1,
                          double th2, double* w, double* res)
{
    if (n_frames < 1 || H < 1 || W < 1 || n_sub < 1) return -1;
    const size_t npx = (size_t)H * W;
    const double lambda = log(ROFF / RON), dts = dt / n_sub;
    for (size_t i = 0; i < npx; i++) { w[i] = 0.5; res[i] = RON / exp(-lambda * (1 - w[i])); }
    for (int f = 0; f + 1 < n_frames; f++) {
        const double* a = imgs + (size_t)f * npx;
        const double* b = a + npx;
        for (size_t i = 0; i < npx; i++) {
            const double v = nsof_ref_frame_drive(a[i], b[i], th1, th2);
            double ww = w[i];
            for (int s = 0; s < n_sub; s++) ww = frame_update(ww, v, dts);
            w[i] = ww;
            res[(size_t)(f + 1) * npx + i] = RON / exp(-lambda * (1 - ww));
        }
    }
    return 0;
}

/* ---- compress_image: imresize(double image, [oh ow], 'lanczos3') of simulationcode_v4_transistor_uav.m:111-121 --
 * MATLAB's imresize is not part of the reference; restated from its published algorithm (imresize.m,
 * `contributions`): antialiased Lanczos-3 (kernel stretched by 1/scale when shrinking), P = ceil(6/scale) + 2
 * taps from floor(u - 3/scale), u = x/scale + 0.5(1 - 1/scale), weights normalised, mirrored indices, axis with the
 * smaller scale first (rows on a tie).  PARITY UNPINNED (no MATLAB/Octave, no stored compressed frames).
 * Scalar loops, one axis at a time; img [h][w] float64. */
static double lanczos3(double x)
{
    const double eps = 2.220446049250313e-16, pi = 3.14159265358979323846;
    double f = (sin(pi * x) * sin(pi * x / 3) + eps) / ((pi * pi * x * x / 3) + eps);
    return fabs(x) < 3 ? f : 0.0;
}

static void resize_axis(const double* in, int h, int w, int axis, int out_len, double* out)
{
    const int in_len = axis == 0 ? h : w;
    const double scale = (double)out_len / in_len;
    double kw = 6.0;
    if (scale < 1) kw /= scale;
    const int P = (int)ceil(kw) + 2;
    const int oh = axis == 0 ? out_len : h, ow = axis == 0 ? w : out_len;
    double* wt = (double*)malloc(sizeof(double) * P);
    int* idx = (int*)malloc(sizeof(int) * P);
    for (int o = 0; o < out_len; o++) {
        const double u = (o + 1) / scale + 0.5 * (1 - 1 / scale);
        const double left = floor(u - kw / 2);
        double sum = 0;
        for (int k = 0; k < P; k++) {
            const double pos = left + k;   /* 1-based source position */
            wt[k] = scale < 1 ? scale * lanczos3(scale * (u - pos)) : lanczos3(u - pos);
            sum += wt[k];
            long q = ((long)pos - 1) % (2L * in_len);
            if (q < 0) q += 2L * in_len;
            idx[k] = (int)(q < in_len ? q : 2L * in_len - 1 - q);
        }
        for (int k = 0; k < P; k++) wt[k] /= sum;
        const int other = axis == 0 ? w : h;
        for (int j = 0; j < other; j++) {
            double acc = 0;
            for (int k = 0; k < P; k++)
                acc += wt[k] * (axis == 0 ? in[(size_t)idx[k] * w + j] : in[(size_t)j * w + idx[k]]);
            if (axis == 0) out[(size_t)o * ow + j] = acc;
            else out[(size_t)j * ow + o] = acc;
        }
    }
    (void)oh;
    free(wt);
    free(idx);
}

int nsof_ref_imresize_lanczos3(const double* img, int h, int w, int oh, int ow, double* out)
{
    if (h < 1 || w < 1 || oh < 1 || ow < 1) return -1;
    const double sh = (double)oh / h, sw = (double)ow / w;
    if (sh <= sw) {
        double* tmp = (double*)malloc(sizeof(double) * (size_t)oh * w);
        if (!tmp) return -2;
        resize_axis(img, h, w, 0, oh, tmp);
        resize_axis(tmp, oh, w, 1, ow, out);
        free(tmp);
    } else {
        double* tmp = (double*)malloc(sizeof(double) * (size_t)h * ow);
        if (!tmp) return -2;
        resize_axis(img, h, w, 1, ow, tmp);
        resize_axis(tmp, h, ow, 0, oh, out);
        free(tmp);
    }
    return 0;
}
