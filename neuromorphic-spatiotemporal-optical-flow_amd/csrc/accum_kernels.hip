// Synaptic accumulator ("memristor array"): HIP kernels + host driver for gfx950.
//
// Replaces /root/reference/eventsim/event_mem_sim.py: update_state (:40-57),
// resistance_exp (:60-63) and the slice loop of simulate (:164-286).
//
// Design (MI355X-first, not a translation of the per-slice NumPy passes):
//  * Pixels are independent and time is serial per pixel, so up to 32 consecutive slices are
//    fused into ONE pass over the state: a scatter kernel ORs "pixel active in slice s" bits
//    into a u32 mask per pixel; the update kernel reads w once, replays the 32 slices in
//    registers and writes w once.  HBM traffic drops from 8 B/px/slice to <= 16/S B/px/slice.
//  * When silent_v lies in the device's dead zone [voff, von] (the default, 0 V) an inactive
//    pixel is a bit-exact no-op (dw/dt = 0, clip is the identity on [0,1]); then only the
//    pixels touched by events are visited (compacted list built by the scatter kernel).
//  * Scheme 2's refractory rule couples consecutive slices through next_ok, so its scatter
//    runs one (tiny) launch per slice in stream order; the state update is still fused.
//  * pow/exp go through double precision so that the float32 result is the correctly
//    rounded one in all but ~1e-9 of cases (the reference's NumPy uses SIMD pow/exp that are
//    themselves 1-4 ulp off libm; tolerances are stated in tests/test_accum_*.py).
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "nsof_internal.h"

namespace {

// event_mem_sim.py:20-34
constexpr float VOFF = (float)-0.2, VON = (float)0.1;
constexpr float KOFF = (float)51.03, KON = (float)-2.91;
constexpr float SON = (float)0.2, SOFF = (float)0.8;
constexpr float BON = (float)-5.12, BOFF = (float)3.10;
constexpr float DT = (float)5e-4;
constexpr double RON = 163305.0, ROFF = 2104377.0;
constexpr long long REFRACTORY_US = 800;
constexpr float WINI = 0.5f;
constexpr int MAX_GROUP = 32;

// x ** b for the float32 state update, evaluated in double and rounded once (NumPy's float32 power is accurate to
// about an ulp; this is correctly rounded except within ~1e-13 of a rounding boundary).  The library log()/exp() spend
// most of their ~150 double-precision instructions on ranges and special cases that cannot occur here
// (x = 1 - w*s lies in [0.2, 1] for a state in [0, 1]); the series below need ~55 and are accurate to 4e-14:
//   log: x = m * 2^e with m in [sqrt(1/2), sqrt(2)), z = (m-1)/(m+1), log m = 2z(1 + z^2/3 + ... + z^14/15)
//   exp: y = k ln2 + r with |r| <= ln2/2, exp r = sum r^n/n! (n <= 13), scaled by 2^k
__device__ __forceinline__ double log_unit_range(double x)
{
    int e;
    double m = frexp(x, &e);                       // m in [0.5, 1)
    if (m < 0.70710678118654752440) { m *= 2.0; e -= 1; }
    const double z = (m - 1.0) / (m + 1.0), w = z * z;
    double p = 1.0 / 15.0;
    p = fma(p, w, 1.0 / 13.0);
    p = fma(p, w, 1.0 / 11.0);
    p = fma(p, w, 1.0 / 9.0);
    p = fma(p, w, 1.0 / 7.0);
    p = fma(p, w, 1.0 / 5.0);
    p = fma(p, w, 1.0 / 3.0);
    p = fma(p, w, 1.0);
    const double de = (double)e;
    return fma(de, 0x1.62e42feep-1, fma(de, 0x1.a39ef35793c76p-33, 2.0 * z * p));   // e * ln2 in two parts
}
__device__ __forceinline__ double exp_small(double y)   // |y| < 700
{
    const double k = rint(y * 1.4426950408889634074);
    const double r = fma(-k, 0x1.a39ef35793c76p-33, fma(-k, 0x1.62e42feep-1, y));
    double p = 1.0 / 6227020800.0;
    p = fma(p, r, 1.0 / 479001600.0);
    p = fma(p, r, 1.0 / 39916800.0);
    p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)k);
}
__device__ __forceinline__ float pow_f32(float x, float b)
{
    // a state outside [0,1] handed to the element-wise entry point can make x <= 0: NumPy gives nan for a negative
    // base, 0 or inf for a zero base
    if (!(x > 0.f)) return x < 0.f ? __builtin_nanf("") : (x == 0.f ? (b > 0.f ? 0.f : __builtin_inff()) : x);
    if (x > 2.f || x < 1e-3f) return (float)exp((double)b * log((double)x));   // far outside the model's range
    return (float)exp_small((double)b * log_unit_range((double)x));
}

__device__ __forceinline__ float update_one(float w, float V)
{
    float dwdt = 0.f;
    if (V < VOFF) {
        const float a = V / VOFF - 1.f;
        const float b = pow_f32(1.f - w * SOFF, BOFF);
        dwdt = KOFF * a * b;
    } else if (V > VON) {
        const float a = V / VON - 1.f;
        const float b = pow_f32(1.f - w * SON, BON);
        dwdt = KON * a * b;
    }
    const float wn = w + dwdt * DT;
    return wn < 0.f ? 0.f : (wn > 1.f ? 1.f : wn);
}

// A slice's voltage is one of two values per run (active / silent), so everything of update_one that depends on V
// alone is evaluated once: the branch taken, k * (V/v0 - 1) (the first product of "k * a * b", same rounding), and the
// (s, b) pair of the power term.  The per-slice step is then branch-free: one pow, two multiplies, one add, the clip.
struct Drive {
    float ka, s, b;   // ka = 0 in the dead zone: dw = 0 * pow(..) = 0, w unchanged, as in update_one
};
__device__ __forceinline__ Drive drive_of(float V)
{
    Drive d;
    if (V < VOFF) { d.ka = KOFF * (V / VOFF - 1.f); d.s = SOFF; d.b = BOFF; }
    else if (V > VON) { d.ka = KON * (V / VON - 1.f); d.s = SON; d.b = BON; }
    else { d.ka = 0.f; d.s = SOFF; d.b = BOFF; }
    return d;
}
__device__ __forceinline__ float update_drive(float w, const Drive& d)
{
    const float wn = w + (d.ka * pow_f32(1.f - w * d.s, d.b)) * DT;
    return wn < 0.f ? 0.f : (wn > 1.f ? 1.f : wn);
}

__device__ __forceinline__ float resistance_one(float w, float neg_lam)
{
    const float e = (float)exp((double)(neg_lam * (1.0f - w)));
    return (float)(RON / (double)e);
}

__global__ __launch_bounds__(256) void k_update_state(const float* __restrict__ w, const float* __restrict__ V,
                                                       float* __restrict__ out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        out[i] = update_one(w[i], V[i]);
}

__global__ __launch_bounds__(256) void k_resistance(const float* __restrict__ w, float* __restrict__ out, size_t n,
                                                     float neg_lam)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        out[i] = resistance_one(w[i], neg_lam);
}

__global__ __launch_bounds__(256) void k_fill(float* __restrict__ p, size_t n, float v)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = v;
}

// The surface as an 8-bit frame, one pixel (modes: see k_surface_gray).
__device__ __forceinline__ uint8_t surface_gray_one(float ww, float neg_lam, int mode)
{
    double g;
    if (mode == 0) {
        const double r = (double)resistance_one(ww, neg_lam);
        g = -3366.0 / log10(1.0 / r) - 306.0;
    } else {
        g = (double)(ww * 255.0f);
    }
    g = g < 0.0 ? 0.0 : (g > 255.0 ? 255.0 : g);   // NaN (I == 1 A exactly) cannot occur for R in [Ron, Roff]
    return (uint8_t)g;
}
// Where the fused dense update leaves the frame of the state it has just written (out == nullptr: nowhere).
struct SurfOut {
    uint8_t* out;
    long long stride;
    int W;
    float neg_lam;
    int mode;
};

// Marks (pixel, slice bit); a pixel's first touch in this group enters the compact list once.  The list's counter is ONE
// word: appended to per lane it serialises every first touch of a group at the L2 atomic unit (~10 ns each: 300 us for the
// 33 k events of a 32-slice group at 1 M events/s, found with rocprofv3 in round 3), so the appends of a wave are
// aggregated -- one atomicAdd of the wave's count, ranks from the ballot.  Call with the whole wave (inactive lanes pass
// active = false); list == nullptr (dense update: no list is read) skips the list entirely.
__device__ __forceinline__ void mark(unsigned* mask, unsigned* list, unsigned* count, unsigned pix, unsigned bit, bool active)
{
    const unsigned old = active ? atomicOr(&mask[pix], bit) : 1u;
    if (!list) return;
    const bool first = active && old == 0;
    const unsigned long long b = __ballot(first);
    if (!b) return;
    const int lane = threadIdx.x & 63, leader = __ffsll((long long)b) - 1;
    unsigned base = 0;
    if (lane == leader) base = atomicAdd(count, (unsigned)__popcll(b));
    base = __shfl(base, leader);
    if (first) list[base + (unsigned)__popcll(b & ((1ull << lane) - 1ull))] = pix;
}

// Scheme 1 (:208-217): every event of the group marks (pixel, its slice).  bounds = event
// indices of the group's slice boundaries (n_sl+1 entries), relative to ev0.
__global__ __launch_bounds__(256) void k_scatter_v1(const short* __restrict__ x, const short* __restrict__ y,
                                                     long long ev0, long long n_ev, const long long* __restrict__ bounds,
                                                     int n_sl, int W, unsigned* mask, unsigned* list, unsigned* count,
                                                     unsigned* mask_hi)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const bool live = i < n_ev;
    const long long e = ev0 + (live ? i : 0);
    int lo = 0, hi = n_sl;  // largest s with bounds[s] <= e
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (bounds[mid] <= e) lo = mid; else hi = mid;
    }
    // slices 32..63 of a group (dense update only: n_sl <= 32 otherwise) go to the second mask word
    mark(lo < 32 ? mask : mask_hi, list, count, (unsigned)y[e] * (unsigned)W + (unsigned)x[e], 1u << (lo & 31), live);
}

// Scheme 2 (:237-269), one slice: eligible <=> next_ok[pix] <= t_first; eligible pixels are
// marked once and get next_ok = t_last + REFRACTORY.  A racing thread of the same slice may
// already see the new next_ok (> t_first) and skip -- the pixel is marked either way.
// pol_sel: -1 = any polarity (magnitude mode), else only events with p == pol_sel.
__global__ __launch_bounds__(256) void k_scatter_v2(const short* __restrict__ x, const short* __restrict__ y,
                                                     const signed char* __restrict__ p, long long ev0, long long n_ev,
                                                     int pol_sel, long long t_first, long long t_next, int W,
                                                     unsigned bit, long long* next_ok, unsigned* mask, unsigned* list,
                                                     unsigned* count)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    bool live = i < n_ev;
    const long long e = ev0 + (live ? i : 0);
    if (live && pol_sel >= 0 && (int)p[e] != pol_sel) live = false;
    const unsigned pix = (unsigned)y[e] * (unsigned)W + (unsigned)x[e];
    bool hit = false;
    if (live) {
        const long long ok = __hip_atomic_load(&next_ok[pix], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ok <= t_first) {
            __hip_atomic_store(&next_ok[pix], t_next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            hit = true;
        }
    }
    mark(mask, list, count, pix, bit, hit);
}

// Fused state update over the touched pixels only (silent_v inside the dead zone).
__global__ __launch_bounds__(256) void k_update_sparse(float* __restrict__ w, unsigned* __restrict__ mask,
                                                        const unsigned* __restrict__ list,
                                                        const unsigned* __restrict__ count, int n_sl, float v_act,
                                                        unsigned* zero_next)
{
    const unsigned n = *count;
    if (zero_next && blockIdx.x == 0 && threadIdx.x == 0) *zero_next = 0;   // the NEXT group's list counter (other parity)
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const unsigned pix = list[i];
        unsigned m = mask[pix];
        mask[pix] = 0;
        float ww = w[pix];
        const Drive da = drive_of(v_act);
        for (int s = 0; s < n_sl; s++, m >>= 1)
            if (m & 1u) ww = update_drive(ww, da);
        w[pix] = ww;
    }
}

// Fused state update over every pixel (silent_v outside the dead zone, or forced dense).
// 4 pixels per thread: one 16-B load/store of w and of the mask per lane.
// SIL_NOOP: silent_v lies in the dead zone [voff, von], where update_state leaves w bit-for-bit unchanged
// (dw = 0, w already inside [0,1]), so only the slices whose bit is set are replayed (in slice order) -- the pass
// is then bound by its one read and one write of the state instead of by 32 no-op evaluations per pixel.
template <bool SIL_NOOP>
__global__ __launch_bounds__(256) void k_update_dense(float* __restrict__ w, unsigned* __restrict__ mask, size_t n4,
                                                       size_t n, int n_sl, float v_act, float v_sil, SurfOut so,
                                                       unsigned* __restrict__ mask_hi)
{
    const Drive da = drive_of(v_act), ds = drive_of(v_sil);
    // m: slices 0..31 of the group, mh: slices 32..63 (groups of up to 64 slices: one pass over the array where a frame
    // interval of 33 slices took two)
    auto replay = [&](float ww, unsigned m, unsigned mh) {
        if (SIL_NOOP) {
            for (; m; m &= m - 1) ww = update_drive(ww, da);
            for (; mh; mh &= mh - 1) ww = update_drive(ww, da);
        } else {
            for (int s = 0; s < n_sl; s++) {
                const bool act = s < 32 ? (m >> s) & 1u : (mh >> (s - 32)) & 1u;
                Drive d;
                d.ka = act ? da.ka : ds.ka;
                d.s = act ? da.s : ds.s;
                d.b = act ? da.b : ds.b;
                ww = update_drive(ww, d);
            }
        }
        return ww;
    };
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        if (4 * i + 3 < n) {
            float4 ww = reinterpret_cast<float4*>(w)[i];
            const uint4 mm = reinterpret_cast<uint4*>(mask)[i];
            if (mm.x | mm.y | mm.z | mm.w) reinterpret_cast<uint4*>(mask)[i] = make_uint4(0, 0, 0, 0);
            uint4 mh = make_uint4(0, 0, 0, 0);
            if (mask_hi) {
                mh = reinterpret_cast<uint4*>(mask_hi)[i];
                if (mh.x | mh.y | mh.z | mh.w) reinterpret_cast<uint4*>(mask_hi)[i] = make_uint4(0, 0, 0, 0);
            }
            ww.x = replay(ww.x, mm.x, mh.x);
            ww.y = replay(ww.y, mm.y, mh.y);
            ww.z = replay(ww.z, mm.z, mh.z);
            ww.w = replay(ww.w, mm.w, mh.w);
            reinterpret_cast<float4*>(w)[i] = ww;
            if (so.out) {   // the frame of the new state: saves the separate surface pass (4 B/px read again + a launch)
                const uint8_t g0 = surface_gray_one(ww.x, so.neg_lam, so.mode), g1 = surface_gray_one(ww.y, so.neg_lam, so.mode);
                const uint8_t g2 = surface_gray_one(ww.z, so.neg_lam, so.mode), g3 = surface_gray_one(ww.w, so.neg_lam, so.mode);
                const size_t px = 4 * i;
                const size_t yy = px / (size_t)so.W, xx = px - yy * (size_t)so.W;
                if (xx + 3 < (size_t)so.W && ((so.stride | (long long)xx) & 3) == 0 && (reinterpret_cast<uintptr_t>(so.out) & 3) == 0) {
                    *reinterpret_cast<unsigned*>(so.out + yy * so.stride + xx) =
                        (unsigned)g0 | ((unsigned)g1 << 8) | ((unsigned)g2 << 16) | ((unsigned)g3 << 24);
                } else {   // a group of 4 that straddles two rows, or an unaligned frame
                    const uint8_t gg[4] = {g0, g1, g2, g3};
                    for (int q = 0; q < 4; q++) {
                        const size_t pq = px + q, yq = pq / (size_t)so.W, xq = pq - yq * (size_t)so.W;
                        so.out[yq * so.stride + xq] = gg[q];
                    }
                }
            }
        } else {
            for (size_t j = 4 * i; j < n; j++) {
                const unsigned m = mask[j], mh = mask_hi ? mask_hi[j] : 0u;
                mask[j] = 0;
                if (mask_hi) mask_hi[j] = 0;
                const float wj = replay(w[j], m, mh);
                w[j] = wj;
                if (so.out) so.out[(j / (size_t)so.W) * so.stride + j % (size_t)so.W] = surface_gray_one(wj, so.neg_lam, so.mode);
            }
        }
    }
}

// ---- scheme 2 without a launch per slice (round 3) ----------------------------------------------------------------------
// The refractory rule (event_mem_sim.py:237-269) looks like a chain over slices -- slice s+1 tests the next_ok that slice s
// wrote -- but the chain is PER PIXEL: a pixel's next_ok depends on that pixel's own earlier events only, and within a slice
// every event of a pixel sees the same next_ok (NumPy reads next_ok[ys, xs] before it writes).  So a group of up to 32
// slices needs ONE scatter, "which slices have an event (of the array's polarity) at this pixel" (bit s of E), and the
// eligibility walk moves into the fused state update: per touched pixel, over the set bits of E in slice order,
//     if next_ok <= t_first[s]:  the pixel is driven in slice s;  next_ok = t_last[s] + REFRACTORY
// with the per-slice constants in a 64-entry table.  No atomics on next_ok, two launches per group and array as in scheme 1.
struct RefrTab {
    long long t_first[32], t_next[32];
};

// E bits of a group: split == 0: every event -> array 0 (magnitude mode); split == 1: p == 1 -> array 0, p == 0 -> array 1
// (:238, :250; other polarity values drive nothing).  bounds as in k_scatter_v1.
__global__ __launch_bounds__(256) void k_scatter_v2g(const short* __restrict__ x, const short* __restrict__ y,
                                                      const signed char* __restrict__ p, long long ev0, long long n_ev,
                                                      const long long* __restrict__ bounds, int n_sl, int W, int split,
                                                      unsigned* mask0, unsigned* list0, unsigned* count0, unsigned* mask1,
                                                      unsigned* list1, unsigned* count1)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const bool live = i < n_ev;
    const long long e = ev0 + (live ? i : 0);
    int arr = 0;
    if (split) {
        const int pv = (int)p[e];
        arr = pv == 1 ? 0 : (pv == 0 ? 1 : -1);
    }
    int lo = 0, hi = n_sl;  // largest s with bounds[s] <= e
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (bounds[mid] <= e) lo = mid; else hi = mid;
    }
    const unsigned pix = (unsigned)y[e] * (unsigned)W + (unsigned)x[e];
    mark(mask0, list0, count0, pix, 1u << lo, live && arr == 0);
    if (split) mark(mask1, list1, count1, pix, 1u << lo, live && arr == 1);
}

// slices with an event -> slices in which the pixel is driven; ok = the pixel's next_ok (updated)
__device__ __forceinline__ unsigned refractory_walk(unsigned e, long long& ok, const long long* tf, const long long* tn)
{
    unsigned m = 0;
    for (; e; e &= e - 1) {
        const int s = __ffs((int)e) - 1;
        if (ok <= tf[s]) {
            m |= 1u << s;
            ok = tn[s];
        }
    }
    return m;
}

__global__ __launch_bounds__(256) void k_update_sparse_v2(float* __restrict__ w, unsigned* __restrict__ mask,
                                                           long long* __restrict__ next_ok, const unsigned* __restrict__ list,
                                                           const unsigned* __restrict__ count, RefrTab tab, float v_act,
                                                           unsigned* zero_next)
{
    __shared__ long long tf[32], tn[32];
    if (threadIdx.x < 32) { tf[threadIdx.x] = tab.t_first[threadIdx.x]; tn[threadIdx.x] = tab.t_next[threadIdx.x]; }
    __syncthreads();
    const unsigned n = *count;
    if (zero_next && blockIdx.x == 0 && threadIdx.x == 0) *zero_next = 0;   // the NEXT group's list counter (other parity)
    const Drive da = drive_of(v_act);
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const unsigned pix = list[i];
        const unsigned e = mask[pix];
        mask[pix] = 0;
        long long ok = next_ok[pix];
        unsigned m = refractory_walk(e, ok, tf, tn);
        if (m) {
            next_ok[pix] = ok;
            float ww = w[pix];
            for (; m; m &= m - 1) ww = update_drive(ww, da);
            w[pix] = ww;
        }
    }
}

template <bool SIL_NOOP>
__global__ __launch_bounds__(256) void k_update_dense_v2(float* __restrict__ w, unsigned* __restrict__ mask,
                                                          long long* __restrict__ next_ok, size_t n4, size_t n, int n_sl,
                                                          RefrTab tab, float v_act, float v_sil)
{
    __shared__ long long tf[32], tn[32];
    if (threadIdx.x < 32) { tf[threadIdx.x] = tab.t_first[threadIdx.x]; tn[threadIdx.x] = tab.t_next[threadIdx.x]; }
    __syncthreads();
    const Drive da = drive_of(v_act), ds = drive_of(v_sil);
    auto one = [&](float ww, unsigned e, size_t pix) {
        unsigned m = 0;
        if (e) {
            long long ok = next_ok[pix];
            m = refractory_walk(e, ok, tf, tn);
            if (m) next_ok[pix] = ok;
        }
        if (SIL_NOOP) {
            for (; m; m &= m - 1) ww = update_drive(ww, da);
        } else {
            for (int s = 0; s < n_sl; s++) {
                const bool act = (m >> s) & 1u;
                Drive d;
                d.ka = act ? da.ka : ds.ka;
                d.s = act ? da.s : ds.s;
                d.b = act ? da.b : ds.b;
                ww = update_drive(ww, d);
            }
        }
        return ww;
    };
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        if (4 * i + 3 < n) {
            float4 ww = reinterpret_cast<float4*>(w)[i];
            const uint4 mm = reinterpret_cast<uint4*>(mask)[i];
            const bool any = (mm.x | mm.y | mm.z | mm.w) != 0;
            if (SIL_NOOP && !any) continue;                       // nothing driven: w unchanged bit for bit
            if (any) reinterpret_cast<uint4*>(mask)[i] = make_uint4(0, 0, 0, 0);
            ww.x = one(ww.x, mm.x, 4 * i);
            ww.y = one(ww.y, mm.y, 4 * i + 1);
            ww.z = one(ww.z, mm.z, 4 * i + 2);
            ww.w = one(ww.w, mm.w, 4 * i + 3);
            reinterpret_cast<float4*>(w)[i] = ww;
        } else {
            for (size_t j = 4 * i; j < n; j++) {
                const unsigned e = mask[j];
                mask[j] = 0;
                w[j] = one(w[j], e, j);
            }
        }
    }
}

// ---- scheme 2 as ONE graph launch per group of slices ---------------------------------------------------------------
// The refractory rule makes every slice's scatter depend on the previous slice's, so a group of 32 slices is 32 (split
// mode: 64) tiny dependent launches + the fused update -- launch overhead, not work, bounds scheme 2.  Here the same
// chain is captured once in a HIP graph and replayed per group (opt-in, NSOF_ACCUM_GRAPH=1: measured slower, see
// accum_advance): the kernels take everything that changes from group to group
// (event range, first / last timestamp of each slice, the group's slice count) from device tables, indexed by a device
// counter that the graph's last node advances.
struct SliceRec {
    long long lo, n, t_first, t_next;   // event range (relative to the staged stream) and the refractory timestamps
};
struct GroupRec {
    int first, g;                       // first staged slice of the group, number of slices (<= 32)
};

__global__ __launch_bounds__(256) void k_scatter_v2_tab(const short* __restrict__ x, const short* __restrict__ y,
                                                         const signed char* __restrict__ p,
                                                         const SliceRec* __restrict__ slices,
                                                         const GroupRec* __restrict__ groups, const int* __restrict__ gi,
                                                         int s, int pol_sel, int W, long long* next_ok, unsigned* mask,
                                                         unsigned* list, unsigned* count)
{
    const GroupRec gr = groups[*gi];
    if (s >= gr.g) return;
    const SliceRec sl = slices[gr.first + s];
    const unsigned bit = 1u << s;
    for (long long i0 = (long long)blockIdx.x * 256; i0 < sl.n; i0 += (long long)gridDim.x * 256) {   // whole waves walk together
        const long long i = i0 + threadIdx.x;
        bool live = i < sl.n;
        const long long e = sl.lo + (live ? i : 0);
        if (live && pol_sel >= 0 && (int)p[e] != pol_sel) live = false;
        const unsigned pix = (unsigned)y[e] * (unsigned)W + (unsigned)x[e];
        bool hit = false;
        if (live) {
            const long long ok = __hip_atomic_load(&next_ok[pix], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (ok <= sl.t_first) {
                __hip_atomic_store(&next_ok[pix], sl.t_next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                hit = true;
            }
        }
        mark(mask, list, count, pix, bit, hit);
    }
}

__global__ __launch_bounds__(256) void k_update_sparse_tab(float* __restrict__ w, unsigned* __restrict__ mask,
                                                            const unsigned* __restrict__ list,
                                                            const unsigned* __restrict__ count,
                                                            const GroupRec* __restrict__ groups, const int* __restrict__ gi,
                                                            float v_act)
{
    const int n_sl = groups[*gi].g;
    const unsigned n = *count;
    const Drive da = drive_of(v_act);
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const unsigned pix = list[i];
        unsigned m = mask[pix];
        mask[pix] = 0;
        float ww = w[pix];
        for (int s = 0; s < n_sl; s++, m >>= 1)
            if (m & 1u) ww = update_drive(ww, da);
        w[pix] = ww;
    }
}

template <bool SIL_NOOP>
__global__ __launch_bounds__(256) void k_update_dense_tab(float* __restrict__ w, unsigned* __restrict__ mask, size_t n,
                                                           const GroupRec* __restrict__ groups, const int* __restrict__ gi,
                                                           float v_act, float v_sil)
{
    const int n_sl = groups[*gi].g;
    const Drive da = drive_of(v_act), ds = drive_of(v_sil);
    for (size_t j = (size_t)blockIdx.x * 256 + threadIdx.x; j < n; j += (size_t)gridDim.x * 256) {
        unsigned m = mask[j];
        if (m) mask[j] = 0;
        float ww = w[j];
        if (SIL_NOOP) {
            for (; m; m &= m - 1) ww = update_drive(ww, da);
        } else {
            for (int s = 0; s < n_sl; s++) {
                const bool act = (m >> s) & 1u;
                Drive d;
                d.ka = act ? da.ka : ds.ka;
                d.s = act ? da.s : ds.s;
                d.b = act ? da.b : ds.b;
                ww = update_drive(ww, d);
            }
        }
        w[j] = ww;
    }
}

__global__ void k_group_done(int* gi, unsigned* count)
{
    *gi += 1;
    count[0] = 0;
    count[1] = 0;
}

// Frame-driven variant (/root/reference/simulation/simulationcode_v4_transistor_uav.m:146-227, 332-347), float64:
// drive voltage from the absolute difference of two compressed frames, then n_sub Euler sub-steps of the same ODE.
__device__ __forceinline__ double frame_update(double w, double V, double dt)
{
    double dwdt = 0.0;
    if (V < -0.2) dwdt = 51.03 * (V / -0.2 - 1) * pow(1 - w * 0.8, 3.10);
    else if (V > 0.1) dwdt = -2.91 * (V / 0.1 - 1) * pow(1 - w * 0.2, -5.12);
    const double nw = w + dwdt * dt;
    return nw < 0 ? 0 : (nw > 1 ? 1 : nw);
}

__global__ __launch_bounds__(64) void k_frame_step(const double* __restrict__ a, const double* __restrict__ b,
                                                    double* __restrict__ w, double* __restrict__ res, size_t n,
                                                    double dts, int n_sub, double th1, double th2, double lambda)
{
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const double d = fabs(a[i] * 256 - b[i] * 256);
    double V = d > th1 ? (d + 4) * 0.75 : (d - 5.5) * 0.6;   // func2 == func3 in the source
    V = V > 0 ? -(0.3 * V + 0) : (V < 0 ? -(3 * V + -3) : 0.0);
    double ww = w[i];
    for (int s = 0; s < n_sub; s++) ww = frame_update(ww, V, dts);
    w[i] = ww;
    res[i] = RON / exp(-lambda * (1 - ww));
}

// Temporal-prior surface as an 8-bit frame.
//   mode 0  the reference's bridge from device state to the gating input, g = uint8(clip(-3366 / log10(I) - 306, 0,
//           255)) with I = V_ds / R, V_ds = 1 V (optical_flow_seg.py:426-431, simulationcode_v4_transistor_uav.m:36),
//           evaluated per pixel in double on R = resistance_exp(w) as float32.  Calibrated for arrays that start at
//           w = 0: the event simulator's initial state w = 0.5 (I = 1.7 uA) already maps to 255.
//   mode 1  build-defined linear map of the state itself, g = uint8(255 * w) (float32 product, truncated): the frame
//           the joined events -> surface -> flow pipeline (BASELINE config 5) hands to the flow stage.
__global__ __launch_bounds__(256) void k_surface_gray(const float* __restrict__ w, uint8_t* __restrict__ out, int W, int H,
                                                       ptrdiff_t stride, float neg_lam, int mode)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= W || y >= H) return;
    out[(ptrdiff_t)y * stride + x] = surface_gray_one(w[(size_t)y * W + x], neg_lam, mode);
}

// ---- surface frames as copy + patch (round 4) ---------------------------------------------------------------------------
// With the silent voltage in the dead zone a pixel without events keeps its state bit for bit, so the 8-bit frame of interval
// k differs from that of interval k-1 only at the pixels the interval's events touch (0.4 % of a 3840x2160 sensor at 1 M
// events/s and 33 ms per frame).  The every-pixel pass moves 17 B/px per interval to find that out; here an interval is
//   A  k_frames_scatter_copy   frames[k] = frames[k-1] (2 B/px) while the interval's events are scattered into the per-pixel
//                              slice masks (one 64-bit word: up to 64 slices) and the compact list of touched pixels
//   B  k_frames_update_patch   the touched pixels replay their slices in order (the update of k_update_sparse), store w and
//                              overwrite their byte of frames[k]
// -- same states, same frames (tests/test_accum_gpu.py::test_run_frames_copy_patch_equals_dense_frames).
__global__ __launch_bounds__(256) void k_frames_scatter_copy(const uint8_t* __restrict__ prev, uint8_t* __restrict__ cur, int W, int H,
                                                              long long row_stride, const short* __restrict__ x,
                                                              const short* __restrict__ y, long long ev0, long long n_ev,
                                                              const long long* __restrict__ bounds, int n_sl,
                                                              unsigned long long* mask64, unsigned* list, unsigned* count)
{
    const long long tid = (long long)blockIdx.x * 256 + threadIdx.x, nthreads = (long long)gridDim.x * 256;
    // the events of the interval FIRST (their atomics' latency then overlaps the copy; whole waves take part: the list append
    // is aggregated per wave)
    const long long n_ev_pad = (n_ev + 63) & ~63ll;
    for (long long i = tid; i < n_ev_pad; i += nthreads) {
        const bool live = i < n_ev;
        const long long e = ev0 + (live ? i : 0);
        int lo = 0, hi = n_sl;  // largest s with bounds[s] <= e
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (bounds[mid] <= e) lo = mid; else hi = mid;
        }
        // ONE 64-bit slice mask per pixel: a pixel enters the list exactly once per interval (its first touch), so no two
        // threads of the update ever hold the same pixel
        const unsigned pix = (unsigned)y[e] * (unsigned)W + (unsigned)x[e];
        const unsigned long long old = live ? atomicOr(&mask64[pix], 1ull << lo) : 1ull;
        const bool first = live && old == 0;
        const unsigned long long b = __ballot(first);
        if (b) {
            const int lane = threadIdx.x & 63, leader = __ffsll((long long)b) - 1;
            unsigned base = 0;
            if (lane == leader) base = atomicAdd(count, (unsigned)__popcll(b));
            base = __shfl(base, leader);
            if (first) list[base + (unsigned)__popcll(b & ((1ull << lane) - 1ull))] = pix;
        }
    }
    // ... then this thread's share of the copy
    if (prev) {
        if (row_stride == W && (((size_t)W * H) & 15) == 0 && ((reinterpret_cast<uintptr_t>(prev) | reinterpret_cast<uintptr_t>(cur)) & 15) == 0) {
            const long long n16 = (long long)W * H / 16;
            typedef unsigned u4v __attribute__((ext_vector_type(4)));
            const u4v* s4 = reinterpret_cast<const u4v*>(prev);
            u4v* d4 = reinterpret_cast<u4v*>(cur);
#ifdef NSOF_ACC_COPY_PLAIN
            for (long long i = tid; i < n16; i += nthreads) d4[i] = s4[i];
#else
            for (long long i = tid; i < n16; i += nthreads) __builtin_nontemporal_store(s4[i], d4 + i);
#endif
        } else {
            const long long n = (long long)W * H;
            for (long long i = tid; i < n; i += nthreads) {
                const long long yy = i / W, xx = i - yy * W;
                cur[yy * row_stride + xx] = prev[yy * row_stride + xx];
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_frames_update_patch(float* __restrict__ w, unsigned long long* __restrict__ mask64,
                                                              const unsigned* __restrict__ list,
                                                              const unsigned* __restrict__ count, float v_act, unsigned* zero_next,
                                                              uint8_t* __restrict__ frame, int W, long long row_stride,
                                                              float neg_lam, int mode)
{
    const unsigned n = *count;
    if (zero_next && blockIdx.x == 0 && threadIdx.x == 0) *zero_next = 0;   // the NEXT interval's list counter (other parity)
    const Drive da = drive_of(v_act);
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const unsigned pix = list[i];
        unsigned long long m = mask64[pix];
        mask64[pix] = 0;
        float ww = w[pix];
        for (; m; m >>= 1)
            if (m & 1ull) ww = update_drive(ww, da);
        w[pix] = ww;
        const unsigned yy = pix / (unsigned)W, xx = pix - yy * (unsigned)W;
        frame[(long long)yy * row_stride + xx] = surface_gray_one(ww, neg_lam, mode);
    }
}

// ---- surface frames by a tile-persistent walk (round 4) -------------------------------------------------------------
// Pixels never interact, so nothing forces an interval to be a launch: a WAVE owns a tile of 1024 consecutive pixels for
// the WHOLE run.  Its state (w, 4 KB), the 8-bit bytes of the current frame (1 KB) and a 64-bit slice mask per pixel (8 KB)
// live in LDS; per interval it ORs the tile's events into the masks, lets one lane per touched pixel replay them
// (atomicExch claims the mask), and streams the tile's 1 KB of frame k out -- frames are WRITE-ONLY (1 B/px per frame, no
// read of the previous frame), the state is read and written once per run:
//   k_tile_bucket   the run's events bucketed by (interval, tile): records of 16 bits (pixel in tile, slice in interval);
//                   the order inside a bucket is irrelevant (scheme 1 applies the same drive once per distinct active slice)
//   k_tile_frames   the walk
// -- two launches per call.
constexpr int TILE_PX = 1024, TILE_SHIFT = 10;

// One workgroup per interval buckets that interval's events by tile: the events of interval k are the contiguous range
// [bounds[k * every], bounds[(k + 1) * every]) of the time-sorted stream and their records fill exactly that range of recs,
// so the counting sort is local -- histogram over the tiles in LDS, exclusive scan, fill through LDS cursors -- and no
// global scan or global atomic is needed.  off[k * ntiles + t] = start of bucket (k, t) in recs; off[n_frames * ntiles] = n_ev.
__global__ __launch_bounds__(1024) void k_tile_bucket(const short* __restrict__ x, const short* __restrict__ y, long long ev0,
                                                       const long long* __restrict__ bounds, int every, int W, unsigned ntiles,
                                                       unsigned* __restrict__ off, unsigned short* __restrict__ recs, int n_frames)
{
    extern __shared__ unsigned s_cnt[];           // [ntiles]
    __shared__ long long s_b[65];                 // the interval's slice bounds (event indices)
    __shared__ unsigned s_part[1024];
    const int k = blockIdx.x, tid = threadIdx.x;
    for (unsigned t = tid; t < ntiles; t += 1024) s_cnt[t] = 0;
    if (tid <= every) s_b[tid] = bounds[(size_t)k * every + tid];
    __syncthreads();
    const long long lo = s_b[0], hi = s_b[every];
    const unsigned base = (unsigned)(lo - ev0);
    auto rec_of = [&](long long e, unsigned& tile) -> unsigned {
        int a = 0, b = every;                     // largest s with s_b[s] <= e
        while (b - a > 1) {
            const int mid = (a + b) >> 1;
            if (s_b[mid] <= e) a = mid; else b = mid;
        }
        const unsigned pix = (unsigned)y[e] * (unsigned)W + (unsigned)x[e];
        tile = pix >> TILE_SHIFT;
        return (pix & (TILE_PX - 1)) | ((unsigned)a << TILE_SHIFT);
    };
    for (long long e = lo + tid; e < hi; e += 1024) {
        unsigned tile;
        (void)rec_of(e, tile);
        atomicAdd(&s_cnt[tile], 1u);
    }
    __syncthreads();
    // exclusive scan over the tiles: thread i owns tiles [i * per, (i + 1) * per)
    const unsigned per = (ntiles + 1023) / 1024, t0 = tid * per, t1 = min(t0 + per, ntiles);
    unsigned sum = 0;
    for (unsigned t = t0; t < t1; t++) sum += s_cnt[t];
    s_part[tid] = sum;
    __syncthreads();
    for (unsigned d = 1; d < 1024; d <<= 1) {
        const unsigned v = tid >= (int)d ? s_part[tid - d] : 0;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    unsigned run = s_part[tid] - sum;
    for (unsigned t = t0; t < t1; t++) {
        const unsigned c = s_cnt[t];
        off[(size_t)k * ntiles + t] = base + run;
        s_cnt[t] = run;                           // the fill pass's cursor
        run += c;
    }
    if (k == n_frames - 1 && tid == 0) off[(size_t)n_frames * ntiles] = (unsigned)(hi - ev0);
    __syncthreads();
    for (long long e = lo + tid; e < hi; e += 1024) {
        unsigned tile;
        const unsigned rec = rec_of(e, tile);
        recs[base + atomicAdd(&s_cnt[tile], 1u)] = (unsigned short)rec;
    }
}

__global__ __launch_bounds__(256) void k_tile_frames(float* __restrict__ w, size_t npx, int W, const unsigned* __restrict__ off,
                                                      const unsigned short* __restrict__ recs, unsigned ntiles, int n_frames,
                                                      float v_act, uint8_t* __restrict__ frames, long long row_stride,
                                                      long long frame_stride, float neg_lam, int mode)
{
    __shared__ unsigned long long s_mask[4][TILE_PX];
    __shared__ __attribute__((aligned(16))) float s_w[4][TILE_PX];
    __shared__ __attribute__((aligned(16))) uint8_t s_b[4][TILE_PX];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned tile = blockIdx.x * 4 + wv;
    if (tile >= ntiles) return;                                   // wave-uniform; no block barrier below
    unsigned long long* ml = s_mask[wv];
    float* wl = s_w[wv];
    uint8_t* bl = s_b[wv];
    const size_t p0 = (size_t)tile * TILE_PX + (size_t)lane * 16;   // this lane's 16 pixels (one row: W % 16 == 0)
    const bool live = p0 < npx;
    const size_t yy = live ? p0 / (size_t)W : 0, xx = live ? p0 - yy * (size_t)W : 0;
    typedef float f4v __attribute__((ext_vector_type(4)));
    typedef unsigned u4v __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int j = 0; j < 4; j++) {
        f4v v = live ? reinterpret_cast<const f4v*>(w + p0)[j] : (f4v){0.f, 0.f, 0.f, 0.f};
        reinterpret_cast<f4v*>(wl + lane * 16)[j] = v;
#pragma unroll
        for (int q = 0; q < 4; q++) bl[lane * 16 + 4 * j + q] = surface_gray_one(v[q], neg_lam, mode);
    }
#pragma unroll
    for (int j = 0; j < 16; j++) ml[lane * 16 + j] = 0ull;
    const Drive da = drive_of(v_act);
    bool dirty = false;
    for (int kb = 0; kb < n_frames; kb += 63) {
        // the bucket bounds of up to 63 intervals of this tile, one per lane (lane i: the start of interval kb + i)
        // (buckets are laid out interval-major: bucket (k, tile) = recs[off[k * ntiles + tile] .. off[k * ntiles + tile + 1]))
        const int kn = min(63, n_frames - kb);
        const unsigned mybeg = lane < kn ? off[(size_t)(kb + lane) * ntiles + tile] : 0u;
        const unsigned myend = lane < kn ? off[(size_t)(kb + lane) * ntiles + tile + 1] : 0u;
        // the first 64 records of the next PF intervals are in flight (a record load is ~1 us of L2 latency against ~0.15 us
        // of work per interval)
        constexpr int PF = 4;
        unsigned rq[PF];
#pragma unroll
        for (int d = 0; d < PF; d++) {
            const unsigned qb = d < kn ? __builtin_amdgcn_readlane(mybeg, d) : 0u, qe = d < kn ? __builtin_amdgcn_readlane(myend, d) : 0u;
            rq[d] = qb + lane < qe ? recs[qb + lane] : 0xffffffffu;
        }
        for (int i0 = 0; i0 < kn; i0 += PF) {
#pragma unroll
          for (int d = 0; d < PF; d++) {
            const int i = i0 + d;
            if (i >= kn) break;                                   // wave-uniform
            const unsigned rcur = rq[d];
            const unsigned c0 = __builtin_amdgcn_readlane(mybeg, i), c1 = __builtin_amdgcn_readlane(myend, i);
            if (i + PF < kn) {
                const unsigned qb = __builtin_amdgcn_readlane(mybeg, i + PF), qe = __builtin_amdgcn_readlane(myend, i + PF);
                rq[d] = qb + lane < qe ? recs[qb + lane] : 0xffffffffu;
            }
            if (c1 > c0) {                                        // wave-uniform
                dirty = true;
                // pass 1: every event of the bucket marks (pixel, slice)
                if (rcur != 0xffffffffu) atomicOr(&ml[rcur & (TILE_PX - 1)], 1ull << (rcur >> TILE_SHIFT));
                for (unsigned q = c0 + 64 + lane; q < c1; q += 64) {
                    const unsigned r = recs[q];
                    atomicOr(&ml[r & (TILE_PX - 1)], 1ull << (r >> TILE_SHIFT));
                }
                // pass 2: one lane per touched pixel claims its mask and replays the slices (LDS operations of a wave
                // execute in order: every mark above is visible here; the fences keep the compiler from moving them)
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                auto settle = [&](unsigned r) {
                    const unsigned pix = r & (TILE_PX - 1);
                    unsigned long long m = atomicExch(&ml[pix], 0ull);
                    if (m) {
                        float ww = wl[pix];
                        for (; m; m &= m - 1) ww = update_drive(ww, da);
                        wl[pix] = ww;
                        bl[pix] = surface_gray_one(ww, neg_lam, mode);
                    }
                };
                if (rcur != 0xffffffffu) settle(rcur);
                for (unsigned q = c0 + 64 + lane; q < c1; q += 64) settle(recs[q]);
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            }
            if (live) {
                const u4v bytes = reinterpret_cast<const u4v*>(bl)[lane];
                __builtin_nontemporal_store(bytes, reinterpret_cast<u4v*>(frames + (size_t)(kb + i) * frame_stride + yy * row_stride + xx));
            }
          }
        }
    }
    if (dirty && live) {
#pragma unroll
        for (int j = 0; j < 4; j++) reinterpret_cast<f4v*>(w + p0)[j] = reinterpret_cast<const f4v*>(wl + lane * 16)[j];
    }
}

// bincount_2d (event_mem_sim.py:100-104): events per pixel.
__global__ __launch_bounds__(256) void k_bincount(const short* __restrict__ x, const short* __restrict__ y, size_t n,
                                                   int W, int* __restrict__ counts)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        atomicAdd(&counts[(size_t)y[i] * W + x[i]], 1);
}

inline int grid_for(size_t n, int cap = 4096)
{
    size_t g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > (size_t)cap ? cap : g));
}

}  // namespace

struct nsof_accum {
    nsof_ctx* ctx = nullptr;
    int H = 0, W = 0, scheme = 1, split = 0;
    float active_v = 0, silent_v = 0;
    int force_dense = 0;   // nsof_accum_set_dense: 0 automatic, 1 every-pixel pass, -1 event-pixel update (where exact)
    size_t npx = 0;
    float* w[2] = {nullptr, nullptr};
    long long* next_ok[2] = {nullptr, nullptr};
    unsigned* mask[2] = {nullptr, nullptr};
    unsigned* mask_hi = nullptr;   // slices 32..63 of a dense scheme-1 group (allocated on first use, kept zero between groups)
    unsigned long long* mask64 = nullptr;   // nsof_accum_run_frames (copy + patch): one 64-bit slice mask per pixel
    // nsof_accum_run_frames (tile walk): bucket counters / offsets [intervals * tiles (+1)] and the 16-bit event records
    unsigned* tile_cnt = nullptr;
    unsigned* tile_off = nullptr;
    unsigned short* tile_recs = nullptr;
    size_t tile_nb_cap = 0, tile_rec_cap = 0;
    unsigned* list[2] = {nullptr, nullptr};
    size_t list_cap = 0;
    unsigned* count = nullptr;  // [2]
    // event staging
    short *dx = nullptr, *dy = nullptr;
    signed char* dp = nullptr;
    long long* dbounds = nullptr;
    size_t ev_cap = 0, bounds_cap = 0;
    // snapshots
    float* snap[2] = {nullptr, nullptr};
    int64_t snap_cap = 0, snap_count = 0;
    int64_t slice_counter = 0;
    // staged stream (nsof_accum_set_events / the staging half of nsof_accum_step_events): slice bounds relative to
    // the first staged event, and for scheme 2 the first / last+refractory timestamp of every slice
    std::vector<long long> h_rel, h_tfirst, h_tnext;
    // scheme 2 graph replay: per-slice / per-group tables on the device, the group counter, the captured graph
    SliceRec* d_slices = nullptr;
    size_t slices_cap = 0;
    GroupRec* d_groups = nullptr;
    size_t groups_cap = 0;
    int* d_gi = nullptr;
    hipGraphExec_t graph = nullptr;
    bool graph_dense = false, graph_sparse_ok = false;
    int use_graph = -1;   // -1: from the environment (NSOF_ACCUM_GRAPH=1 switches it on), 0 / 1: forced
    int v2_per_slice = -1;   // scheme 2: 0 = one scatter per group + refractory walk in the update (default), 1 = one scatter per slice
    int frames_path = 0;     // nsof_accum_run_frames: 0 = the tile walk where it applies, 1 = copy + patch per interval (kept as the cross-check)
};

static int accum_alloc(nsof_ctx* ctx, void** p, size_t bytes)
{
    hipError_t e = hipMalloc(p, bytes ? bytes : 1);
    if (e != hipSuccess) {
        *p = nullptr;
        return nsof_set_error(ctx, NSOF_ENOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    }
    return NSOF_OK;
}

extern "C" void nsof_accum_destroy(nsof_accum* a)
{
    if (!a) return;
    hipSetDevice(a->ctx->device);
    hipStreamSynchronize(a->ctx->stream);
    for (int i = 0; i < 2; i++) {
        hipFree(a->w[i]); hipFree(a->next_ok[i]); hipFree(a->mask[i]); hipFree(a->list[i]); hipFree(a->snap[i]);
    }
    hipFree(a->mask_hi); hipFree(a->mask64); hipFree(a->tile_cnt); hipFree(a->tile_off); hipFree(a->tile_recs); hipFree(a->count); hipFree(a->dx); hipFree(a->dy); hipFree(a->dp); hipFree(a->dbounds);
    hipFree(a->d_slices); hipFree(a->d_groups); hipFree(a->d_gi);
    if (a->graph) hipGraphExecDestroy(a->graph);
    delete a;
}

extern "C" int nsof_accum_reset(nsof_accum* a)
{
    if (!a) return NSOF_EINVAL;
    nsof_ctx* ctx = a->ctx;
    NSOF_HIP(ctx, hipSetDevice(ctx->device));
    const int narr = a->split ? 2 : 1;
    for (int i = 0; i < narr; i++) {
        hipLaunchKernelGGL(k_fill, dim3(grid_for(a->npx)), dim3(256), 0, ctx->stream, a->w[i], a->npx, WINI);
        NSOF_HIP(ctx, hipMemsetAsync(a->mask[i], 0, a->npx * sizeof(unsigned), ctx->stream));
        if (a->scheme == 2) NSOF_HIP(ctx, hipMemsetAsync(a->next_ok[i], 0, a->npx * sizeof(long long), ctx->stream));
    }
    NSOF_HIP(ctx, hipGetLastError());
    a->slice_counter = 0;
    a->snap_count = 0;
    return NSOF_OK;
}

extern "C" int nsof_accum_create(nsof_ctx* ctx, int height, int width, int scheme, int polarity_split, float active_v,
                                 float silent_v, nsof_accum** out)
{
    if (!ctx || !out) return NSOF_EINVAL;
    *out = nullptr;
    if (height < 1 || width < 1 || (scheme != 1 && scheme != 2))
        return nsof_set_error(ctx, NSOF_EINVAL, "bad accumulator geometry %dx%d or scheme %d", height, width, scheme);
    if ((size_t)height * width > 0xFFFFFFFFull) return nsof_set_error(ctx, NSOF_EUNSUPPORTED, "sensor too large");
    NSOF_HIP(ctx, hipSetDevice(ctx->device));
    nsof_accum* a = new (std::nothrow) nsof_accum();
    if (!a) return nsof_set_error(ctx, NSOF_ENOMEM, "out of host memory");
    a->ctx = ctx; a->H = height; a->W = width; a->scheme = scheme;
    a->split = (scheme == 2 && polarity_split) ? 1 : 0;
    a->active_v = active_v; a->silent_v = silent_v;
    a->npx = (size_t)height * width;
    const int narr = a->split ? 2 : 1;
    int rc = accum_alloc(ctx, (void**)&a->count, 4 * sizeof(unsigned));   // [parity][array]: groups alternate, see accum_advance
    for (int i = 0; i < narr && !rc; i++) {
        rc = accum_alloc(ctx, (void**)&a->w[i], (a->npx + 4) * sizeof(float));
        if (!rc) rc = accum_alloc(ctx, (void**)&a->mask[i], (a->npx + 4) * sizeof(unsigned));
        if (!rc && scheme == 2) rc = accum_alloc(ctx, (void**)&a->next_ok[i], a->npx * sizeof(long long));
    }
    if (!rc) rc = nsof_accum_reset(a);
    if (rc) { nsof_accum_destroy(a); return rc; }
    *out = a;
    return NSOF_OK;
}

extern "C" int nsof_accum_set_frames_path(nsof_accum* a, int path)
{
    if (!a || path < 0 || path > 1) return NSOF_EINVAL;
    a->frames_path = path;
    return NSOF_OK;
}

extern "C" int nsof_accum_set_dense(nsof_accum* a, int force_dense)
{
    if (!a) return NSOF_EINVAL;
    a->force_dense = force_dense > 0 ? 1 : (force_dense < 0 ? -1 : 0);
    return NSOF_OK;
}

static int accum_snapshot(nsof_accum* a)
{
    nsof_ctx* ctx = a->ctx;
    const int narr = a->split ? 2 : 1;
    if (a->snap_count == a->snap_cap) {
        const int64_t ncap = a->snap_cap ? a->snap_cap * 2 : 16;
        for (int i = 0; i < narr; i++) {
            float* nb = nullptr;
            int rc = accum_alloc(ctx, (void**)&nb, (size_t)ncap * a->npx * sizeof(float));
            if (rc) return rc;
            if (a->snap_count)
                NSOF_HIP(ctx, hipMemcpyAsync(nb, a->snap[i], (size_t)a->snap_count * a->npx * sizeof(float),
                                             hipMemcpyDeviceToDevice, ctx->stream));
            NSOF_HIP(ctx, hipStreamSynchronize(ctx->stream));
            hipFree(a->snap[i]);
            a->snap[i] = nb;
        }
        a->snap_cap = ncap;
    }
    const float neg_lam = (float)(-std::log(ROFF / RON));
    for (int i = 0; i < narr; i++)
        hipLaunchKernelGGL(k_resistance, dim3(grid_for(a->npx)), dim3(256), 0, ctx->stream, a->w[i],
                           a->snap[i] + (size_t)a->snap_count * a->npx, a->npx, neg_lam);
    NSOF_HIP(ctx, hipGetLastError());
    a->snap_count++;
    return NSOF_OK;
}

// Upload the events of slices [0, n_slices) (bounds sb index the caller's arrays) and keep what the slice loop needs
// on the host: the bounds relative to the first uploaded event and, for scheme 2, every slice's first timestamp and
// last timestamp + refractory period.
static int accum_stage(nsof_accum* a, const int16_t* x, const int16_t* y, const int8_t* p, const int64_t* t,
                       const int64_t* sb, int64_t n_slices)
{
    nsof_ctx* ctx = a->ctx;
    if (n_slices < 0 || !sb) return nsof_set_error(ctx, NSOF_EINVAL, "bad slice bounds");
    NSOF_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t e0 = sb[0], e1 = sb[n_slices], n_ev = e1 - e0;
    if (n_ev < 0) return nsof_set_error(ctx, NSOF_EINVAL, "slice bounds not monotone");
    if (n_ev > 0 && (!x || !y || !t || (a->scheme == 2 && a->split && !p)))
        return nsof_set_error(ctx, NSOF_EINVAL, "null event array");
    for (int64_t s = 0; s < n_slices; s++)
        if (sb[s + 1] < sb[s]) return nsof_set_error(ctx, NSOF_EINVAL, "slice bounds not monotone");
    // validate coordinates on the host: an out-of-range event would be an out-of-bounds store on the device
    for (int64_t e = e0; e < e1; e++)
        if ((unsigned)x[e] >= (unsigned)a->W || (unsigned)y[e] >= (unsigned)a->H)
            return nsof_set_error(ctx, NSOF_EINVAL, "event %lld at (%d,%d) outside the %dx%d sensor", (long long)e,
                                  (int)x[e], (int)y[e], a->W, a->H);
    const int narr = a->split ? 2 : 1;
    int rc;
    if ((size_t)n_ev > a->ev_cap) {
        NSOF_HIP(ctx, hipStreamSynchronize(ctx->stream));
        hipFree(a->dx); hipFree(a->dy); hipFree(a->dp);
        for (int i = 0; i < 2; i++) { hipFree(a->list[i]); a->list[i] = nullptr; }
        a->dx = a->dy = nullptr; a->dp = nullptr;
        const size_t cap = (size_t)n_ev + (size_t)n_ev / 4 + 1024;
        if ((rc = accum_alloc(ctx, (void**)&a->dx, cap * 2)) || (rc = accum_alloc(ctx, (void**)&a->dy, cap * 2)) ||
            (rc = accum_alloc(ctx, (void**)&a->dp, cap)))
            return rc;
        for (int i = 0; i < narr; i++)
            if ((rc = accum_alloc(ctx, (void**)&a->list[i], cap * sizeof(unsigned)))) return rc;
        a->ev_cap = cap;
        if (a->graph) { hipGraphExecDestroy(a->graph); a->graph = nullptr; }   // captured pointers are stale
    }
    if ((size_t)(n_slices + 1) > a->bounds_cap) {
        NSOF_HIP(ctx, hipStreamSynchronize(ctx->stream));
        hipFree(a->dbounds);
        if ((rc = accum_alloc(ctx, (void**)&a->dbounds, (size_t)(n_slices + 1) * 8))) return rc;
        a->bounds_cap = (size_t)(n_slices + 1);
    }
    if (n_ev > 0) {
        NSOF_HIP(ctx, hipMemcpyAsync(a->dx, x + e0, (size_t)n_ev * 2, hipMemcpyHostToDevice, ctx->stream));
        NSOF_HIP(ctx, hipMemcpyAsync(a->dy, y + e0, (size_t)n_ev * 2, hipMemcpyHostToDevice, ctx->stream));
        if (p) NSOF_HIP(ctx, hipMemcpyAsync(a->dp, p + e0, (size_t)n_ev, hipMemcpyHostToDevice, ctx->stream));
    }
    a->h_rel.resize((size_t)n_slices + 1);
    for (int64_t s = 0; s <= n_slices; s++) a->h_rel[s] = (long long)(sb[s] - e0);
    a->h_tfirst.assign((size_t)n_slices, 0);
    a->h_tnext.assign((size_t)n_slices, 0);
    if (a->scheme == 2)
        for (int64_t s = 0; s < n_slices; s++)
            if (sb[s + 1] > sb[s]) {
                a->h_tfirst[s] = t[sb[s]];
                a->h_tnext[s] = t[sb[s + 1] - 1] + REFRACTORY_US;
            }
    NSOF_HIP(ctx, hipMemcpyAsync(a->dbounds, a->h_rel.data(), a->h_rel.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    std::vector<SliceRec> recs;
    if (a->scheme == 2 && n_slices > 0) {   // per-slice table of the graph replay
        recs.resize((size_t)n_slices);
        for (int64_t s = 0; s < n_slices; s++)
            recs[s] = SliceRec{a->h_rel[s], a->h_rel[s + 1] - a->h_rel[s], a->h_tfirst[s], a->h_tnext[s]};
        if ((size_t)n_slices > a->slices_cap) {
            NSOF_HIP(ctx, hipStreamSynchronize(ctx->stream));
            hipFree(a->d_slices);
            a->d_slices = nullptr;
            if ((rc = accum_alloc(ctx, (void**)&a->d_slices, (size_t)n_slices * sizeof(SliceRec)))) return rc;
            a->slices_cap = (size_t)n_slices;
            if (a->graph) { hipGraphExecDestroy(a->graph); a->graph = nullptr; }   // captured pointers are stale
        }
        NSOF_HIP(ctx, hipMemcpyAsync(a->d_slices, recs.data(), recs.size() * sizeof(SliceRec), hipMemcpyHostToDevice, ctx->stream));
    }
    NSOF_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the caller's arrays (and recs) are not retained
    return NSOF_OK;
}

// Scheme 2 reads two timestamps per slice -- its first event's (the refractory test, event_mem_sim.py:243,253,265) and
// its last event's (+ REFRACTORY_US -> next_ok, :246,256,267) -- and they belong to the slice of the WHOLE stream.  A
// row band (nsof.dist.simulate_banded) stages only its own events, whose first / last differ: the caller hands the
// global table over, and the band's state then equals its rows of the unsharded run.
extern "C" int nsof_accum_set_slice_times(nsof_accum* a, const int64_t* t_first, const int64_t* t_last, int64_t n_slices)
{
    if (!a) return NSOF_EINVAL;
    nsof_ctx* ctx = a->ctx;
    if (!t_first || !t_last || n_slices != (int64_t)a->h_tfirst.size())
        return nsof_set_error(ctx, NSOF_EINVAL, "set_slice_times: %lld slices given, %zu staged", (long long)n_slices, a->h_tfirst.size());
    if (a->scheme != 2) return NSOF_OK;   // scheme 1 reads no timestamps
    NSOF_HIP(ctx, hipSetDevice(ctx->device));
    for (int64_t s = 0; s < n_slices; s++) {
        a->h_tfirst[s] = t_first[s];
        a->h_tnext[s] = t_last[s] + REFRACTORY_US;
    }
    if (n_slices > 0 && a->d_slices) {
        std::vector<SliceRec> recs((size_t)n_slices);
        for (int64_t s = 0; s < n_slices; s++)
            recs[s] = SliceRec{a->h_rel[s], a->h_rel[s + 1] - a->h_rel[s], a->h_tfirst[s], a->h_tnext[s]};
        NSOF_HIP(ctx, hipMemcpyAsync(a->d_slices, recs.data(), recs.size() * sizeof(SliceRec), hipMemcpyHostToDevice, ctx->stream));
        NSOF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return NSOF_OK;
}

// Advance over staged slices [s_begin, s_begin + n_slices).
static int accum_surface(nsof_accum* a, int which, const SurfOut& so)
{
    nsof_ctx* ctx = a->ctx;
    dim3 grid((a->W + 255) / 256, a->H);
    hipLaunchKernelGGL(k_surface_gray, grid, dim3(256), 0, ctx->stream, a->w[which], so.out, a->W, a->H, (ptrdiff_t)so.stride,
                       so.neg_lam, so.mode);
    NSOF_HIP(ctx, hipGetLastError());
    return NSOF_OK;
}

// surf (optional): after the LAST slice of the call the surface of array surf_which goes to surf->out as an 8-bit frame --
// fused into the last group's dense scheme-1 update where that kernel runs, a separate k_surface_gray launch otherwise.
static int accum_advance(nsof_accum* a, int64_t s_begin, int64_t n_slices, int64_t snap_every, const SurfOut* surf = nullptr,
                         int surf_which = 0)
{
    nsof_ctx* ctx = a->ctx;
    if (s_begin < 0 || n_slices < 0 || (size_t)(s_begin + n_slices + 1) > a->h_rel.size())
        return nsof_set_error(ctx, NSOF_EINVAL, "slices [%lld, %lld) outside the staged stream", (long long)s_begin,
                              (long long)(s_begin + n_slices));
    NSOF_HIP(ctx, hipSetDevice(ctx->device));
    const int narr = a->split ? 2 : 1;
    int rc;
    const std::vector<long long>& rel = a->h_rel;
    const bool dead_zone = !(a->silent_v < VOFF) && !(a->silent_v > VON);
    // Scheme 1 with the silent voltage in the dead zone: the event-pixel update (lists, groups of 32 slices) and the
    // every-pixel pass (groups of 64 slices, no lists) give the same bits; which one is faster depends on the sensor size.
    // Measured with a frame every 33 slices (scripts/accum_mode_probe.py): 1280x720 0.39 vs 0.75-0.89 ms per 30 frames,
    // 3840x2160 0.97 vs 1.21 ms for the every-pixel pass -- both are launch bound and it needs half the launches; its cost
    // grows with the pixel count (16 B/px per 64 slices), so beyond ~12 M pixels the event-pixel update takes over.
    static const size_t auto_dense_px = [] {
        const char* e = NSOF_AB_GETENV("NSOF_ACCUM_AUTO_DENSE_PX");
        return e ? (size_t)atoll(e) : (size_t)12 << 20;
    }();
    const bool sparse = dead_zone && a->force_dense <= 0 && (a->force_dense < 0 || !(a->scheme == 1 && a->npx <= auto_dense_px));
    const float v_act = a->scheme == 1 ? a->active_v : a->silent_v + a->active_v;

    int64_t s0 = s_begin;
    const int64_t s_end = s_begin + n_slices;
    bool surf_done = false;
    if (a->use_graph < 0) {
        // measured (scripts/bench_accum_v2.py, 3840x2160, 1 M events/s): the replayed graph is 7-12 % SLOWER than the
        // plain launches (split 130 k vs 140 k slices/s, magnitude 224 k vs 255 k) -- a graph node costs as much as a
        // stream launch here -- so it is opt-in
        const char* e = NSOF_AB_GETENV("NSOF_ACCUM_GRAPH");
        a->use_graph = (e && e[0] == '1') ? 1 : 0;
    }
    if (a->v2_per_slice < 0) {   // NSOF_ACCUM_V2=slices: round 2's one-scatter-launch-per-slice form (A/B runs)
        const char* e = NSOF_AB_GETENV("NSOF_ACCUM_V2");
        a->v2_per_slice = (e && e[0] == 's') ? 1 : 0;
    }
    if (a->use_graph) a->v2_per_slice = 1;   // the graph replays the per-slice chain
#ifdef NSOF_AB   // tuning builds only: a group's per-slice chain replayed as a HIP graph (measured 7-12 % slower than plain launches)
    if (a->scheme == 2 && a->use_graph && n_slices > 0) {
        // groups of this call (same rule as below: up to 32 slices, ending right after a snapshot slice)
        std::vector<GroupRec> groups;
        std::vector<char> snap_after;
        {
            int64_t c = a->slice_counter, q = s_begin;
            while (q < s_end) {
                int64_t g = s_end - q < MAX_GROUP ? s_end - q : MAX_GROUP;
                if (snap_every > 0) {
                    const int64_t to_snap = (c % snap_every == 0) ? 1 : (snap_every - c % snap_every) + 1;
                    if (to_snap < g) g = to_snap;
                }
                groups.push_back(GroupRec{(int)q, (int)g});
                c += g;
                q += g;
                snap_after.push_back(snap_every > 0 && (c - 1) % snap_every == 0);
            }
        }
        if (groups.size() > a->groups_cap) {
            NSOF_HIP(ctx, hipStreamSynchronize(ctx->stream));
            hipFree(a->d_groups);
            a->d_groups = nullptr;
            const size_t cap = groups.size() + groups.size() / 2 + 16;
            if ((rc = accum_alloc(ctx, (void**)&a->d_groups, cap * sizeof(GroupRec)))) return rc;
            a->groups_cap = cap;
            if (a->graph) { hipGraphExecDestroy(a->graph); a->graph = nullptr; }
        }
        if (!a->d_gi && (rc = accum_alloc(ctx, (void**)&a->d_gi, sizeof(int)))) return rc;
        NSOF_HIP(ctx, hipMemcpyAsync(a->d_groups, groups.data(), groups.size() * sizeof(GroupRec), hipMemcpyHostToDevice, ctx->stream));
        NSOF_HIP(ctx, hipMemsetAsync(a->d_gi, 0, sizeof(int), ctx->stream));
        NSOF_HIP(ctx, hipMemsetAsync(a->count, 0, 2 * sizeof(unsigned), ctx->stream));
        NSOF_HIP(ctx, hipStreamSynchronize(ctx->stream));   // `groups` is a stack-lifetime buffer
        if (a->graph && (a->graph_dense != !sparse)) { hipGraphExecDestroy(a->graph); a->graph = nullptr; }
        if (!a->graph) {
            hipGraph_t gr = nullptr;
            NSOF_HIP(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
            const dim3 sg(16), blk(256);
            for (int s = 0; s < MAX_GROUP; s++) {
                if (a->split) {
                    hipLaunchKernelGGL(k_scatter_v2_tab, sg, blk, 0, ctx->stream, a->dx, a->dy, a->dp, a->d_slices, a->d_groups,
                                       a->d_gi, s, 1, a->W, a->next_ok[0], a->mask[0], a->list[0], a->count);
                    hipLaunchKernelGGL(k_scatter_v2_tab, sg, blk, 0, ctx->stream, a->dx, a->dy, a->dp, a->d_slices, a->d_groups,
                                       a->d_gi, s, 0, a->W, a->next_ok[1], a->mask[1], a->list[1], a->count + 1);
                } else {
                    hipLaunchKernelGGL(k_scatter_v2_tab, sg, blk, 0, ctx->stream, a->dx, a->dy, a->dp, a->d_slices, a->d_groups,
                                       a->d_gi, s, -1, a->W, a->next_ok[0], a->mask[0], a->list[0], a->count);
                }
            }
            for (int i = 0; i < narr; i++) {
                if (sparse)
                    hipLaunchKernelGGL(k_update_sparse_tab, dim3(64), blk, 0, ctx->stream, a->w[i], a->mask[i], a->list[i],
                                       a->count + i, a->d_groups, a->d_gi, v_act);
                else if (dead_zone)
                    hipLaunchKernelGGL(k_update_dense_tab<true>, dim3(grid_for(a->npx, 8192)), blk, 0, ctx->stream, a->w[i],
                                       a->mask[i], a->npx, a->d_groups, a->d_gi, v_act, a->silent_v);
                else
                    hipLaunchKernelGGL(k_update_dense_tab<false>, dim3(grid_for(a->npx, 8192)), blk, 0, ctx->stream, a->w[i],
                                       a->mask[i], a->npx, a->d_groups, a->d_gi, v_act, a->silent_v);
            }
            hipLaunchKernelGGL(k_group_done, dim3(1), dim3(1), 0, ctx->stream, a->d_gi, a->count);
            hipError_t e1 = hipStreamEndCapture(ctx->stream, &gr);
            if (e1 != hipSuccess) return nsof_set_error(ctx, NSOF_EDEVICE, "graph capture failed: %s", hipGetErrorString(e1));
            hipError_t e2 = hipGraphInstantiate(&a->graph, gr, nullptr, nullptr, 0);
            hipGraphDestroy(gr);
            if (e2 != hipSuccess) { a->graph = nullptr; return nsof_set_error(ctx, NSOF_EDEVICE, "graph instantiate failed: %s", hipGetErrorString(e2)); }
            a->graph_dense = !sparse;
        }
        for (size_t gi = 0; gi < groups.size(); gi++) {
            {
                nsof_prof_scope ps(ctx, NSOF_K_ACCUM);
                NSOF_HIP(ctx, hipGraphLaunch(a->graph, ctx->stream));
            }
            a->slice_counter += groups[gi].g;
            if (snap_after[gi])
                if ((rc = accum_snapshot(a))) return rc;
        }
        return surf ? accum_surface(a, surf_which, *surf) : NSOF_OK;
    }
#endif
    // the dense scheme-1 update takes groups of up to 64 slices (two mask words per pixel): a 33-slice frame interval is one
    // pass over the array instead of two
    const bool wide = a->scheme == 1 && !sparse;
    if (wide && !a->mask_hi) {
        if ((rc = accum_alloc(ctx, (void**)&a->mask_hi, a->npx * sizeof(unsigned) + 16))) return rc;
        NSOF_HIP(ctx, hipMemsetAsync(a->mask_hi, 0, a->npx * sizeof(unsigned) + 16, ctx->stream));
    }
    const int64_t max_group = wide ? 2 * MAX_GROUP : MAX_GROUP;
    // List counters of the event-pixel update: two sets used alternately.  A group's update kernels zero the OTHER set -- the
    // one the next group's scatter appends to -- so the per-group 8-byte memset (a launch of its own) is gone: one per call.
    int par = 0;
    if (sparse && s0 < s_end) NSOF_HIP(ctx, hipMemsetAsync(a->count, 0, 4 * sizeof(unsigned), ctx->stream));
    while (s0 < s_end) {
        // group = up to max_group slices, ending right after the next snapshot slice
        int64_t g = s_end - s0 < max_group ? s_end - s0 : max_group;
        if (snap_every > 0) {
            const int64_t c = a->slice_counter;
            const int64_t to_snap = (c % snap_every == 0) ? 1 : (snap_every - c % snap_every) + 1;
            if (to_snap < g) g = to_snap;
        }
        const long long ge0 = rel[s0], ge1 = rel[s0 + g], gn = ge1 - ge0;
        unsigned* const cnt = a->count + 2 * par;          // this group's counters; (the dense update has no list)
        unsigned* const cnt_next = a->count + 2 * (par ^ 1);
        if (sparse && gn == 0) NSOF_HIP(ctx, hipMemsetAsync(cnt_next, 0, 2 * sizeof(unsigned), ctx->stream));   // no update kernel will
        unsigned* const l0 = sparse ? a->list[0] : nullptr;                      // the dense update reads no list
        unsigned* const l1 = sparse ? a->list[a->split ? 1 : 0] : nullptr;
        if (gn > 0) {
            nsof_prof_scope ps(ctx, NSOF_K_ACCUM);
            if (a->scheme == 1) {
                hipLaunchKernelGGL(k_scatter_v1, dim3((unsigned)((gn + 255) / 256)), dim3(256), 0, ctx->stream, a->dx,
                                   a->dy, ge0, gn, a->dbounds + s0, (int)g, a->W, a->mask[0], l0, cnt, a->mask_hi);
            } else if (!a->v2_per_slice) {
                hipLaunchKernelGGL(k_scatter_v2g, dim3((unsigned)((gn + 255) / 256)), dim3(256), 0, ctx->stream, a->dx, a->dy,
                                   a->dp, ge0, gn, a->dbounds + s0, (int)g, a->W, a->split ? 1 : 0, a->mask[0], l0,
                                   cnt, a->mask[a->split ? 1 : 0], l1, cnt + 1);
            } else {
                for (int64_t s = 0; s < g; s++) {
                    const long long lo = rel[s0 + s], hi = rel[s0 + s + 1];
                    if (hi <= lo) continue;
                    const long long t_first = a->h_tfirst[s0 + s], t_next = a->h_tnext[s0 + s];
                    const unsigned bit = 1u << s;
                    const dim3 grid((unsigned)((hi - lo + 255) / 256));
                    if (a->split) {
                        hipLaunchKernelGGL(k_scatter_v2, grid, dim3(256), 0, ctx->stream, a->dx, a->dy, a->dp, lo,
                                           hi - lo, 1, t_first, t_next, a->W, bit, a->next_ok[0], a->mask[0],
                                           l0, cnt);
                        hipLaunchKernelGGL(k_scatter_v2, grid, dim3(256), 0, ctx->stream, a->dx, a->dy, a->dp, lo,
                                           hi - lo, 0, t_first, t_next, a->W, bit, a->next_ok[1], a->mask[1],
                                           l1, cnt + 1);
                    } else {
                        hipLaunchKernelGGL(k_scatter_v2, grid, dim3(256), 0, ctx->stream, a->dx, a->dy, a->dp, lo,
                                           hi - lo, -1, t_first, t_next, a->W, bit, a->next_ok[0], a->mask[0],
                                           l0, cnt);
                    }
                }
            }
            NSOF_HIP(ctx, hipGetLastError());
        }
        {
            nsof_prof_scope ps(ctx, NSOF_K_ACCUM);
            RefrTab tab;
            const bool refr = a->scheme == 2 && !a->v2_per_slice;
            if (refr)
                for (int s = 0; s < 32; s++) {
                    const bool live = s < g && rel[s0 + s + 1] > rel[s0 + s];
                    tab.t_first[s] = live ? a->h_tfirst[s0 + s] : 0;
                    tab.t_next[s] = live ? a->h_tnext[s0 + s] : 0;
                }
            for (int i = 0; i < narr; i++) {
                if (refr) {
                    const size_t n4 = (a->npx + 3) / 4;
                    if (sparse) {
                        if (gn > 0)
                            hipLaunchKernelGGL(k_update_sparse_v2, dim3(grid_for((size_t)gn, 1024)), dim3(256), 0, ctx->stream,
                                               a->w[i], a->mask[i], a->next_ok[i], a->list[i], cnt + i, tab, v_act, cnt_next + i);
                    } else if (dead_zone) {
                        hipLaunchKernelGGL(k_update_dense_v2<true>, dim3(grid_for(n4, 8192)), dim3(256), 0, ctx->stream, a->w[i],
                                           a->mask[i], a->next_ok[i], n4, a->npx, (int)g, tab, v_act, a->silent_v);
                    } else {
                        hipLaunchKernelGGL(k_update_dense_v2<false>, dim3(grid_for(n4, 8192)), dim3(256), 0, ctx->stream, a->w[i],
                                           a->mask[i], a->next_ok[i], n4, a->npx, (int)g, tab, v_act, a->silent_v);
                    }
                } else if (sparse) {
                    if (gn > 0)
                        hipLaunchKernelGGL(k_update_sparse, dim3(grid_for((size_t)gn, 1024)), dim3(256), 0, ctx->stream,
                                           a->w[i], a->mask[i], a->list[i], cnt + i, (int)g, v_act, cnt_next + i);
                } else {
                    const size_t n4 = (a->npx + 3) / 4;
                    SurfOut so{nullptr, 0, a->W, 0.f, 0};
                    if (surf && i == surf_which && s0 + g == s_end) {   // the call's last group: leave the frame as well
                        so = *surf;
                        surf_done = true;
                    }
                    if (dead_zone)
                        hipLaunchKernelGGL(k_update_dense<true>, dim3(grid_for(n4, 8192)), dim3(256), 0, ctx->stream,
                                           a->w[i], a->mask[i], n4, a->npx, (int)g, v_act, a->silent_v, so,
                                           g > MAX_GROUP ? a->mask_hi : nullptr);
                    else
                        hipLaunchKernelGGL(k_update_dense<false>, dim3(grid_for(n4, 8192)), dim3(256), 0, ctx->stream,
                                           a->w[i], a->mask[i], n4, a->npx, (int)g, v_act, a->silent_v, so,
                                           g > MAX_GROUP ? a->mask_hi : nullptr);
                }
            }
            NSOF_HIP(ctx, hipGetLastError());
        }
        a->slice_counter += g;
        s0 += g;
        if (sparse) par ^= 1;
        if (snap_every > 0 && (a->slice_counter - 1) % snap_every == 0)
            if ((rc = accum_snapshot(a))) return rc;
    }
    return surf && !surf_done ? accum_surface(a, surf_which, *surf) : NSOF_OK;
}

extern "C" int nsof_accum_step_events(nsof_accum* a, const int16_t* x, const int16_t* y, const int8_t* p,
                                      const int64_t* t, const int64_t* sb, int64_t n_slices, int64_t snap_every)
{
    if (!a) return NSOF_EINVAL;
    if (n_slices == 0) return NSOF_OK;
    if (int rc = accum_stage(a, x, y, p, t, sb, n_slices)) return rc;
    return accum_advance(a, 0, n_slices, snap_every);
}

extern "C" int nsof_accum_set_events(nsof_accum* a, const int16_t* x, const int16_t* y, const int8_t* p,
                                     const int64_t* t, const int64_t* sb, int64_t n_slices)
{
    if (!a) return NSOF_EINVAL;
    return accum_stage(a, x, y, p, t, sb, n_slices);
}

extern "C" int nsof_accum_run(nsof_accum* a, int64_t first_slice, int64_t n_slices, int64_t snap_every)
{
    if (!a) return NSOF_EINVAL;
    return accum_advance(a, first_slice, n_slices, snap_every);
}

extern "C" int nsof_accum_surface_u8_dev(nsof_accum* a, int which, int mode, uint8_t* d_out, ptrdiff_t row_stride)
{
    if (!a || !d_out || which < 0 || which > (a->split ? 1 : 0) || row_stride < a->W || mode < 0 || mode > 1) return NSOF_EINVAL;
    NSOF_HIP(a->ctx, hipSetDevice(a->ctx->device));
    return accum_surface(a, which, SurfOut{d_out, (long long)row_stride, a->W, (float)(-std::log(ROFF / RON)), mode});
}

extern "C" int nsof_accum_run_surface(nsof_accum* a, int64_t first_slice, int64_t n_slices, int which, int mode, uint8_t* d_out,
                                      ptrdiff_t row_stride)
{
    if (!a || !d_out || which < 0 || which > (a->split ? 1 : 0) || row_stride < a->W || mode < 0 || mode > 1) return NSOF_EINVAL;
    const SurfOut so{d_out, (long long)row_stride, a->W, (float)(-std::log(ROFF / RON)), mode};
    return accum_advance(a, first_slice, n_slices, 0, &so, which);
}

// n_frames consecutive intervals of `every` slices, the surface after each into d_frames[k] (k-th frame at + k * frame_stride
// bytes).  Scheme 1 with the silent voltage in the dead zone and no forced every-pixel pass: frames as copy + patch (above),
// two small launches per interval issued from this one call; otherwise n_frames x nsof_accum_run_surface.
extern "C" int nsof_accum_run_frames(nsof_accum* a, int64_t first_slice, int64_t n_frames, int64_t every, int which, int mode,
                                     uint8_t* d_frames, ptrdiff_t row_stride, ptrdiff_t frame_stride)
{
    if (!a || !d_frames || which < 0 || which > (a->split ? 1 : 0) || row_stride < a->W || mode < 0 || mode > 1 || n_frames < 0 ||
        every < 1 || frame_stride < 0)
        return NSOF_EINVAL;
    nsof_ctx* ctx = a->ctx;
    if (n_frames == 0) return NSOF_OK;
    if (first_slice < 0 || (size_t)(first_slice + n_frames * every + 1) > a->h_rel.size())
        return nsof_set_error(ctx, NSOF_EINVAL, "slices [%lld, %lld) outside the staged stream", (long long)first_slice,
                              (long long)(first_slice + n_frames * every));
    const bool dead_zone = !(a->silent_v < VOFF) && !(a->silent_v > VON);
    const float neg_lam = (float)(-std::log(ROFF / RON));
    if (!(a->scheme == 1 && dead_zone && a->force_dense <= 0 && every <= 2 * MAX_GROUP)) {
        for (int64_t k = 0; k < n_frames; k++) {
            const SurfOut so{d_frames + k * frame_stride, (long long)row_stride, a->W, neg_lam, mode};
            if (int rc = accum_advance(a, first_slice + k * every, every, 0, &so, which)) return rc;
        }
        return NSOF_OK;
    }
    NSOF_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    const std::vector<long long>& relv = a->h_rel;
    const long long ev0 = relv[first_slice], n_ev = relv[first_slice + n_frames * every] - ev0;
    // ---- the tile walk: whole run in four launches (frames write-only); needs 16-byte-addressable frame rows
    const bool tile_ok = (a->W & 15) == 0 && (row_stride & 15) == 0 && (frame_stride & 15) == 0 &&
                         (reinterpret_cast<uintptr_t>(d_frames) & 15) == 0 && n_frames >= 2 && n_ev < (1ll << 31) &&
                         (a->npx + TILE_PX - 1) / TILE_PX <= 15000 &&   // the bucketing workgroup's histogram: 4 B per tile of LDS
                         (size_t)n_frames * ((a->npx + TILE_PX - 1) / TILE_PX) < ((size_t)1 << 30) && a->frames_path != 1;
    if (tile_ok) {
        const unsigned ntiles = (unsigned)((a->npx + TILE_PX - 1) / TILE_PX);
        const size_t nb = (size_t)n_frames * ntiles;
        if (nb + 1 > a->tile_nb_cap) {
            NSOF_HIP(ctx, hipStreamSynchronize(ctx->stream));
            hipFree(a->tile_cnt); hipFree(a->tile_off);
            a->tile_cnt = a->tile_off = nullptr;
            a->tile_nb_cap = 0;
            const size_t cap = nb + nb / 4 + 1024;
            if ((rc = accum_alloc(ctx, (void**)&a->tile_off, cap * 4))) return rc;
            a->tile_nb_cap = cap;
        }
        if ((size_t)n_ev > a->tile_rec_cap) {
            NSOF_HIP(ctx, hipStreamSynchronize(ctx->stream));
            hipFree(a->tile_recs);
            a->tile_recs = nullptr;
            a->tile_rec_cap = 0;
            const size_t cap = (size_t)n_ev + (size_t)n_ev / 4 + 1024;
            if ((rc = accum_alloc(ctx, (void**)&a->tile_recs, cap * 2))) return rc;
            a->tile_rec_cap = cap;
        }
        nsof_prof_scope ps(ctx, NSOF_K_ACCUM);
        hipLaunchKernelGGL(k_tile_bucket, dim3((unsigned)n_frames), dim3(1024), (size_t)ntiles * 4, ctx->stream, a->dx, a->dy, ev0,
                           a->dbounds + first_slice, (int)every, a->W, ntiles, a->tile_off, a->tile_recs, (int)n_frames);
        hipLaunchKernelGGL(k_tile_frames, dim3((ntiles + 3) / 4), dim3(256), 0, ctx->stream, a->w[0], a->npx, a->W,
                           (const unsigned*)a->tile_off, (const unsigned short*)a->tile_recs, ntiles, (int)n_frames, a->active_v, d_frames,
                           (long long)row_stride, (long long)frame_stride, neg_lam, mode);
        NSOF_HIP(ctx, hipGetLastError());
        a->slice_counter += n_frames * every;
        return NSOF_OK;
    }
    if (!a->mask64) {   // per-pixel 64-bit slice masks of this path (kept zero between intervals)
        if ((rc = accum_alloc(ctx, (void**)&a->mask64, a->npx * sizeof(unsigned long long)))) return rc;
        NSOF_HIP(ctx, hipMemsetAsync(a->mask64, 0, a->npx * sizeof(unsigned long long), ctx->stream));
    }
    const std::vector<long long>& rel = a->h_rel;
    NSOF_HIP(ctx, hipMemsetAsync(a->count, 0, 4 * sizeof(unsigned), ctx->stream));
    int par = 0;
#ifndef NSOF_ACC_COPY_BLOCKS
#define NSOF_ACC_COPY_BLOCKS 4096
#endif
    const unsigned copy_blocks = (unsigned)std::min<size_t>(NSOF_ACC_COPY_BLOCKS, (a->npx / 16 + 255) / 256 + 1);
    for (int64_t k = 0; k < n_frames; k++) {
        const int64_t s0 = first_slice + k * every;
        const long long ge0 = rel[s0], gn = rel[s0 + every] - ge0;
        unsigned* const cnt = a->count + 2 * par;
        unsigned* const cnt_next = a->count + 2 * (par ^ 1);
        uint8_t* const cur = d_frames + k * frame_stride;
        const uint8_t* const prev = k > 0 ? d_frames + (k - 1) * frame_stride : nullptr;
        nsof_prof_scope ps(ctx, NSOF_K_ACCUM);
        if (prev || gn > 0) {
            const unsigned blocks = prev ? copy_blocks : (unsigned)((gn + 255) / 256);
            hipLaunchKernelGGL(k_frames_scatter_copy, dim3(blocks), dim3(256), 0, ctx->stream, prev, cur, a->W, a->H,
                               (long long)row_stride, a->dx, a->dy, ge0, gn, a->dbounds + s0, (int)every, a->mask64, a->list[0], cnt);
        }
        // (launched for an empty interval as well: it zeroes the next interval's counter)
        hipLaunchKernelGGL(k_frames_update_patch, dim3(grid_for((size_t)std::max<long long>(gn, 1), 1024)), dim3(256), 0, ctx->stream,
                           a->w[0], a->mask64, a->list[0], cnt, a->active_v, cnt_next, cur, a->W,
                           (long long)row_stride, neg_lam, mode);
        NSOF_HIP(ctx, hipGetLastError());
        if (k == 0) {   // the call's first frame has no predecessor to copy: one pass over the array
            const SurfOut so{cur, (long long)row_stride, a->W, neg_lam, mode};
            if ((rc = accum_surface(a, which, so))) return rc;
        }
        a->slice_counter += every;
        par ^= 1;
    }
    return NSOF_OK;
}

// Checkpoint / resume: the whole state of one array is w (float32 [H][W]), its refractory map (int64 [H][W],
// scheme 2) and the global slice counter that times the snapshots.
extern "C" int nsof_accum_read_state(nsof_accum* a, int which, float* w_out, int64_t* next_ok_out, int64_t* slice_counter)
{
    if (!a || which < 0 || which > (a->split ? 1 : 0)) return NSOF_EINVAL;
    nsof_ctx* ctx = a->ctx;
    NSOF_HIP(ctx, hipSetDevice(ctx->device));
    if (w_out) NSOF_HIP(ctx, hipMemcpyAsync(w_out, a->w[which], a->npx * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    if (next_ok_out) {
        if (a->next_ok[which])
            NSOF_HIP(ctx, hipMemcpyAsync(next_ok_out, a->next_ok[which], a->npx * 8, hipMemcpyDeviceToHost, ctx->stream));
        else
            memset(next_ok_out, 0, a->npx * 8);
    }
    NSOF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (slice_counter) *slice_counter = a->slice_counter;
    return NSOF_OK;
}

extern "C" int nsof_accum_write_state(nsof_accum* a, int which, const float* w_in, const int64_t* next_ok_in,
                                      int64_t slice_counter)
{
    if (!a || which < 0 || which > (a->split ? 1 : 0)) return NSOF_EINVAL;
    nsof_ctx* ctx = a->ctx;
    if (slice_counter < 0) return nsof_set_error(ctx, NSOF_EINVAL, "negative slice counter");
    NSOF_HIP(ctx, hipSetDevice(ctx->device));
    if (w_in) NSOF_HIP(ctx, hipMemcpyAsync(a->w[which], w_in, a->npx * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    if (next_ok_in && a->next_ok[which])
        NSOF_HIP(ctx, hipMemcpyAsync(a->next_ok[which], next_ok_in, a->npx * 8, hipMemcpyHostToDevice, ctx->stream));
    NSOF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    a->slice_counter = slice_counter;
    return NSOF_OK;
}

extern "C" int nsof_accum_update_state_dev(nsof_ctx* ctx, const float* d_w, const float* d_V, float* d_out, size_t n)
{
    if (!ctx || !d_w || !d_V || !d_out) return NSOF_EINVAL;
    if (!n) return NSOF_OK;
    nsof_prof_scope ps(ctx, NSOF_K_ACCUM);
    hipLaunchKernelGGL(k_update_state, dim3(grid_for(n)), dim3(256), 0, ctx->stream, d_w, d_V, d_out, n);
    NSOF_HIP(ctx, hipGetLastError());
    return NSOF_OK;
}

// Host arrays in, host histogram out (the reference calls it on the events of one slice).
extern "C" int nsof_accum_bincount_2d(nsof_ctx* ctx, const int16_t* x, const int16_t* y, size_t n, int height,
                                      int width, int32_t* counts_out)
{
    if (!ctx) return NSOF_EINVAL;
    if (!counts_out || (n && (!x || !y))) return nsof_set_error(ctx, NSOF_EINVAL, "null pointer");
    if (height < 1 || width < 1) return nsof_set_error(ctx, NSOF_ESHAPE, "empty sensor");
    for (size_t i = 0; i < n; i++)   // np.bincount raises on negative values; out-of-range ones would grow the array
        if (x[i] < 0 || y[i] < 0 || x[i] >= width || y[i] >= height)
            return nsof_set_error(ctx, NSOF_EINVAL, "event %zu (%d,%d) outside the %dx%d sensor", i, (int)x[i],
                                  (int)y[i], width, height);
    NSOF_HIP(ctx, hipSetDevice(ctx->device));
    const size_t npx = (size_t)height * width;
    const size_t szE = (n * 2 + 255) & ~(size_t)255, szC = (npx * 4 + 255) & ~(size_t)255;
    int rc = nsof_ws_reserve(ctx, &ctx->stage, &ctx->stage_bytes, 2 * szE + szC);
    if (rc) return rc;
    short* dx = (short*)ctx->stage;
    short* dy = (short*)((char*)ctx->stage + szE);
    int* dc = (int*)((char*)ctx->stage + 2 * szE);
    NSOF_HIP(ctx, hipMemsetAsync(dc, 0, npx * 4, ctx->stream));
    if (n) {
        NSOF_HIP(ctx, hipMemcpyAsync(dx, x, n * 2, hipMemcpyHostToDevice, ctx->stream));
        NSOF_HIP(ctx, hipMemcpyAsync(dy, y, n * 2, hipMemcpyHostToDevice, ctx->stream));
        nsof_prof_scope ps(ctx, NSOF_K_ACCUM);
        hipLaunchKernelGGL(k_bincount, dim3(grid_for(n)), dim3(256), 0, ctx->stream, dx, dy, n, width, dc);
    }
    NSOF_HIP(ctx, hipGetLastError());
    NSOF_HIP(ctx, hipMemcpyAsync(counts_out, dc, npx * 4, hipMemcpyDeviceToHost, ctx->stream));
    NSOF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return NSOF_OK;
}

extern "C" int nsof_accum_resistance_dev(nsof_ctx* ctx, const float* d_w, float* d_out, size_t n)
{
    if (!ctx || !d_w || !d_out) return NSOF_EINVAL;
    if (!n) return NSOF_OK;
    hipLaunchKernelGGL(k_resistance, dim3(grid_for(n)), dim3(256), 0, ctx->stream, d_w, d_out, n,
                       (float)(-std::log(ROFF / RON)));
    NSOF_HIP(ctx, hipGetLastError());
    return NSOF_OK;
}

extern "C" int nsof_accum_read_w(nsof_accum* a, int which, float* out)
{
    if (!a || !out || which < 0 || which > (a->split ? 1 : 0)) return NSOF_EINVAL;
    nsof_ctx* ctx = a->ctx;
    NSOF_HIP(ctx, hipMemcpyAsync(out, a->w[which], a->npx * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    NSOF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return NSOF_OK;
}

extern "C" int nsof_accum_read_resistance(nsof_accum* a, int which, float* out)
{
    if (!a || !out || which < 0 || which > (a->split ? 1 : 0)) return NSOF_EINVAL;
    nsof_ctx* ctx = a->ctx;
    float* tmp = nullptr;
    int rc = accum_alloc(ctx, (void**)&tmp, a->npx * sizeof(float));
    if (rc) return rc;
    rc = nsof_accum_resistance_dev(ctx, a->w[which], tmp, a->npx);
    if (!rc) {
        hipError_t e = hipMemcpyAsync(out, tmp, a->npx * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = nsof_set_error(ctx, NSOF_EDEVICE, "copy failed: %s", hipGetErrorString(e));
    }
    hipFree(tmp);
    return rc;
}

extern "C" int64_t nsof_accum_snapshot_count(const nsof_accum* a) { return a ? a->snap_count : 0; }

// Block maximum of the device current I = v_ds / R over memsize x memsize pixel blocks: the gating image's input
// (one value per block) formed on the device instead of from a downloaded surface.
// max(v_ds / R) = v_ds / min(R) exactly (division by a positive float is monotone), so a block reduces min(R) in
// float32 -- R from a stored snapshot, or resistance_one(w) of the current state -- and thread 0 divides in double.
__global__ __launch_bounds__(256) void k_block_min_resistance(const float* __restrict__ src, int is_w, int W, int memsize,
                                                               int cols, float neg_lam, double v_ds, double* __restrict__ out)
{
    __shared__ float part[4];
    const int bx = blockIdx.x, by = blockIdx.y;
    const float* base = src + ((size_t)by * memsize) * W + (size_t)bx * memsize;
    float m = INFINITY;
    for (int i = threadIdx.x; i < memsize * memsize; i += 256) {
        const float v = base[(size_t)(i / memsize) * W + (i % memsize)];
        m = fminf(m, is_w ? resistance_one(v, neg_lam) : v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fminf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) out[(size_t)by * cols + bx] = v_ds / (double)fminf(fminf(part[0], part[1]), fminf(part[2], part[3]));
}

extern "C" int nsof_accum_block_current(nsof_accum* a, int which, int64_t snapshot, int memsize, double v_ds, double* out)
{
    if (!a || !out || which < 0 || which > (a->split ? 1 : 0) || memsize < 1 || memsize > a->W || memsize > a->H || !(v_ds > 0))
        return NSOF_EINVAL;
    if (snapshot >= a->snap_count) return nsof_set_error(a->ctx, NSOF_EINVAL, "snapshot %lld of %lld", (long long)snapshot, (long long)a->snap_count);
    nsof_ctx* ctx = a->ctx;
    NSOF_HIP(ctx, hipSetDevice(ctx->device));
    const int rows = a->H / memsize, cols = a->W / memsize;
    const size_t bytes = sizeof(double) * rows * cols;
    int rc = nsof_ws_reserve(ctx, &ctx->tmp, &ctx->tmp_bytes, bytes);
    if (rc) return rc;
    const float* src = snapshot < 0 ? a->w[which] : a->snap[which] + (size_t)snapshot * a->npx;
    hipLaunchKernelGGL(k_block_min_resistance, dim3(cols, rows), dim3(256), 0, ctx->stream, src, snapshot < 0 ? 1 : 0, a->W,
                       memsize, cols, (float)(-std::log(ROFF / RON)), v_ds, (double*)ctx->tmp);
    NSOF_HIP(ctx, hipGetLastError());
    NSOF_HIP(ctx, hipMemcpyAsync(out, ctx->tmp, bytes, hipMemcpyDeviceToHost, ctx->stream));
    NSOF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return NSOF_OK;
}

// Device twin: the map goes to DEVICE memory (d_out, rows x cols doubles), nothing is synchronised -- the input of
// nsof_roi_from_surface_dev, so that events -> surface -> ROI rectangles never visits the host.
extern "C" int nsof_accum_block_current_dev(nsof_accum* a, int which, int64_t snapshot, int memsize, double v_ds, double* d_out)
{
    if (!a || !d_out || which < 0 || which > (a->split ? 1 : 0) || memsize < 1 || memsize > a->W || memsize > a->H || !(v_ds > 0))
        return NSOF_EINVAL;
    if (snapshot >= a->snap_count) return nsof_set_error(a->ctx, NSOF_EINVAL, "snapshot %lld of %lld", (long long)snapshot, (long long)a->snap_count);
    nsof_ctx* ctx = a->ctx;
    NSOF_HIP(ctx, hipSetDevice(ctx->device));
    const int rows = a->H / memsize, cols = a->W / memsize;
    const float* src = snapshot < 0 ? a->w[which] : a->snap[which] + (size_t)snapshot * a->npx;
    hipLaunchKernelGGL(k_block_min_resistance, dim3(cols, rows), dim3(256), 0, ctx->stream, src, snapshot < 0 ? 1 : 0, a->W,
                       memsize, cols, (float)(-std::log(ROFF / RON)), v_ds, d_out);
    NSOF_HIP(ctx, hipGetLastError());
    return NSOF_OK;
}

extern "C" int nsof_accum_read_snapshots(nsof_accum* a, int which, float* out, int64_t max_count)
{
    if (!a || which < 0 || which > (a->split ? 1 : 0)) return NSOF_EINVAL;
    nsof_ctx* ctx = a->ctx;
    const int64_t n = a->snap_count < max_count ? a->snap_count : max_count;
    if (n > 0) {
        if (!out) return NSOF_EINVAL;
        NSOF_HIP(ctx, hipMemcpyAsync(out, a->snap[which], (size_t)n * a->npx * sizeof(float), hipMemcpyDeviceToHost,
                                     ctx->stream));
        NSOF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (which == (a->split ? 1 : 0)) a->snap_count = 0;  // cleared after the last array has been read
    return NSOF_OK;
}

extern "C" int64_t nsof_accum_slice_bounds(const int64_t* t, int64_t n, int64_t slice_us, int64_t* idx, int64_t cap)
{
    if (!t || n <= 0 || slice_us <= 0) return 0;
    const int64_t start = t[0], stop = t[n - 1] + slice_us;
    int64_t nb = (stop - start + slice_us - 1) / slice_us;
    if (nb < 0) nb = 0;
    if (idx) {
        int64_t pos = 0;
        for (int64_t i = 0; i < nb && i < cap; i++) {
            const int64_t b = start + i * slice_us;
            while (pos < n && t[pos] < b) pos++;
            idx[i] = pos;
        }
    }
    return nb;
}

// Frame-driven accumulator on HOST arrays: imgs float64 [n_frames][H][W] (compressed frames in [0,1]);
// w_out float64 [H][W]; res_out float64 [n_frames][H][W] (initial state, then one snapshot per frame pair).
extern "C" int nsof_accum_frames_f64(nsof_ctx* ctx, const double* imgs, int n_frames, int height, int width, double dt,
                                     int n_sub_steps, double th1, double th2, double* w_out, double* res_out)
{
    if (!ctx) return NSOF_EINVAL;
    if (!imgs || !w_out || !res_out || n_frames < 1 || height < 1 || width < 1 || n_sub_steps < 1)
        return nsof_set_error(ctx, NSOF_EINVAL, "bad frame-accumulator arguments");
    NSOF_HIP(ctx, hipSetDevice(ctx->device));
    const size_t npx = (size_t)height * width;
    double *d_img = nullptr, *d_w = nullptr, *d_res = nullptr;
    int rc = accum_alloc(ctx, (void**)&d_img, (size_t)n_frames * npx * 8);
    if (!rc) rc = accum_alloc(ctx, (void**)&d_w, npx * 8);
    if (!rc) rc = accum_alloc(ctx, (void**)&d_res, (size_t)n_frames * npx * 8);
    if (!rc) {
        const double lambda = std::log(ROFF / RON);
        std::vector<double> init(npx, 0.5), r0(npx, RON / std::exp(-lambda * 0.5));
        hipError_t e = hipMemcpyAsync(d_img, imgs, (size_t)n_frames * npx * 8, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_w, init.data(), npx * 8, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_res, r0.data(), npx * 8, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        for (int f = 0; f + 1 < n_frames && e == hipSuccess; f++) {
            hipLaunchKernelGGL(k_frame_step, dim3((unsigned)((npx + 63) / 64)), dim3(64), 0, ctx->stream,
                               d_img + (size_t)f * npx, d_img + (size_t)(f + 1) * npx, d_w, d_res + (size_t)(f + 1) * npx,
                               npx, dt / n_sub_steps, n_sub_steps, th1, th2, lambda);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(w_out, d_w, npx * 8, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess)
            e = hipMemcpyAsync(res_out, d_res, (size_t)n_frames * npx * 8, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = nsof_set_error(ctx, NSOF_EDEVICE, "frame accumulator: %s", hipGetErrorString(e));
    }
    hipFree(d_img); hipFree(d_w); hipFree(d_res);
    return rc;
}
