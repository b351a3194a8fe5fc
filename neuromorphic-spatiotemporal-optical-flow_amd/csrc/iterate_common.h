// Device helpers shared by the fused Farneback iteration kernels (farneback_iterate.hip, farneback_iterate_x.hip):
// the expansion layout, the per-row load bundle of a pixel, FarnebackUpdateMatrices for one pixel, the flow source.
#pragma once
#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "nsof_internal.h"

namespace {

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int floor_f(float v)
{
    int i = (int)v;
    return i - (i > v);
}

// R of one image: [h][w][4] f32 (channels 0-3 of a pixel = one aligned 16-B access) followed by [h][w] f32
// (channel 4).  32-bit byte offsets against wave-uniform bases keep every load in the "SGPR base + VGPR offset"
// form.  The L1 serves 4 lanes per cycle whatever the access width, so a thread-row costs 9 loads here
// (R0: x4 + x1; R1: 2 rows x (x4, x4, x2)) instead of 15 with planar channels.
struct Planes {
    const char* q4;   // interleaved channels 0..3
    const char* c4;   // channel 4
};
struct __attribute__((packed, aligned(4))) f2u {  // two adjacent floats, only 4-byte aligned
    float a, b;
};
__device__ __forceinline__ Planes planes_of(const float* img_base, size_t plane)
{
    Planes p;
    p.q4 = reinterpret_cast<const char*>(img_base);
    p.c4 = reinterpret_cast<const char*>(img_base + 4 * plane);
    return p;
}

struct RowIn {
    float4 z;             // R0 channels 0..3
    float z4;             // R0 channel 4
    float4 t0, t1, b0, b1;  // R1 channels 0..3 at (y1,x1), (y1,x1+1), (y1+1,x1), (y1+1,x1+1)
    f2u t4, b4;           // R1 channel 4 at (y1, x1..x1+1) and (y1+1, x1..x1+1)
    float dx, dy, fx, fy;
    int inside;
};

// Issue every load one (row, column) needs; `d` is the flow at that pixel (already loaded).
__device__ __forceinline__ void issue_row(RowIn& in, const Planes& R0, const Planes& R1, int W, int H, int x, int y,
                                          float2 d)
{
    const unsigned pix = (unsigned)y * (unsigned)W + (unsigned)x;
    in.dx = d.x;
    in.dy = d.y;
    float fx = x + d.x, fy = y + d.y;
    const int x1 = floor_f(fx), y1 = floor_f(fy);
    in.fx = fx - x1;
    in.fy = fy - y1;
    in.inside = (unsigned)x1 < (unsigned)(W - 1) && (unsigned)y1 < (unsigned)(H - 1);
#if defined(NSOF_ABL) && NSOF_ABL == 3   // timing-only build: no R0 loads either
    in.z = make_float4(d.x, d.y, d.x + 1.f, (float)pix);
    in.z4 = d.y + 2.f;
#else
    in.z = *reinterpret_cast<const float4*>(R0.q4 + pix * 16u);
    in.z4 = *reinterpret_cast<const float*>(R0.c4 + pix * 4u);
#endif
    // The R1 gather is issued unconditionally, at a clamped (always valid) address when the sample falls
    // outside: a load under a lane-dependent branch cannot be counted by s_waitcnt vmcnt(N), which would
    // force every wait down to "almost nothing outstanding" and serialise the software pipeline.
    const int xs = clampi(x1, 0, W - 2), ys = clampi(y1, 0, H - 2);
    const unsigned o = (unsigned)ys * (unsigned)W + (unsigned)xs;
#if defined(NSOF_ABL) && (NSOF_ABL == 1 || NSOF_ABL == 3)   // timing-only build: no R1 gather
    in.t0 = in.t1 = in.b0 = in.b1 = make_float4(in.z.x + (float)o, in.z.y, in.z.z, in.z.w);
    in.t4.a = in.t4.b = in.b4.a = in.b4.b = in.z4;
#else
    in.t0 = *reinterpret_cast<const float4*>(R1.q4 + o * 16u);
    in.t1 = *reinterpret_cast<const float4*>(R1.q4 + o * 16u + 16u);
    in.b0 = *reinterpret_cast<const float4*>(R1.q4 + (o + (unsigned)W) * 16u);
    in.b1 = *reinterpret_cast<const float4*>(R1.q4 + (o + (unsigned)W) * 16u + 16u);
    in.t4 = *reinterpret_cast<const f2u*>(R1.c4 + o * 4u);
    in.b4 = *reinterpret_cast<const f2u*>(R1.c4 + (o + (unsigned)W) * 4u);
#endif
}

// FarnebackUpdateMatrices for one pixel, from loaded inputs.
__device__ __forceinline__ void matrix_from(const RowIn& in, int x, int y, int W, int H, float (&M)[5])
{
    float r2, r3, r4, r5, r6;
    const float dx = in.dx, dy = in.dy;
    if (in.inside) {
        const float fx = in.fx, fy = in.fy;
        const float a00 = (1.f - fx) * (1.f - fy), a01 = fx * (1.f - fy), a10 = (1.f - fx) * fy, a11 = fx * fy;
        r2 = a00 * in.t0.x + a01 * in.t1.x + a10 * in.b0.x + a11 * in.b1.x;
        r3 = a00 * in.t0.y + a01 * in.t1.y + a10 * in.b0.y + a11 * in.b1.y;
        r4 = a00 * in.t0.z + a01 * in.t1.z + a10 * in.b0.z + a11 * in.b1.z;
        r5 = a00 * in.t0.w + a01 * in.t1.w + a10 * in.b0.w + a11 * in.b1.w;
        r6 = a00 * in.t4.a + a01 * in.t4.b + a10 * in.b4.a + a11 * in.b4.b;
        r4 = (in.z.z + r4) * 0.5f;
        r5 = (in.z.w + r5) * 0.5f;
        r6 = (in.z4 + r6) * 0.25f;
    } else {
        r2 = r3 = 0.f;
        r4 = in.z.z;
        r5 = in.z.w;
        r6 = in.z4 * 0.5f;
    }
    r2 = (in.z.x - r2) * 0.5f;
    r3 = (in.z.y - r3) * 0.5f;
    r2 += r4 * dy + r6 * dx;
    r3 += r6 * dy + r5 * dx;
    if ((unsigned)(x - 5) >= (unsigned)(W - 10) || (unsigned)(y - 5) >= (unsigned)(H - 10)) {
        auto bw = [](int i) { return i < 2 ? 0.14f : 0.4472f; };
        const float scale = (x < 5 ? bw(x) : 1.f) * (x >= W - 5 ? bw(W - x - 1) : 1.f) * (y < 5 ? bw(y) : 1.f) *
                            (y >= H - 5 ? bw(H - y - 1) : 1.f);
        r2 *= scale; r3 *= scale; r4 *= scale; r5 *= scale; r6 *= scale;
    }
    M[0] = r4 * r4 + r6 * r6;
    M[1] = (r4 + r5) * r6;
    M[2] = r5 * r5 + r6 * r6;
    M[3] = r4 * r2 + r6 * r3;
    M[4] = r6 * r2 + r5 * r3;
}

// Where a thread's flow_in comes from.  UPS = false: the level's own flow buffer.  UPS = true (first iteration of
// every level but the coarsest): the previous, coarser level's flow is resampled on the fly with exactly the
// arithmetic of k_flow_upsample (resize INTER_LINEAR, then "flow *= 1/pyr_scale") -- the full-resolution initial
// flow is then never written to or read from HBM.
__device__ __forceinline__ void lin_x(int d, double scale, int slen, int& s, float& a)
{
    float f = (float)((d + 0.5) * scale - 0.5);
    s = floor_f(f);
    a = f - s;
    if (s < 0) { s = 0; a = 0.f; }
    if (s >= slen - 1) { s = slen - 1; a = 0.f; }
}
template <bool UPS>
struct FlowSrc;
// fetch() only issues loads (the result is consumed windows later); resolve() turns what was fetched into the flow.
template <>
struct FlowSrc<false> {
    const char* base;   // flow_in of this pair
    unsigned W, xc;
    using Raw = float2;
    __device__ __forceinline__ Raw fetch(int r) const
    {
        return *reinterpret_cast<const float2*>(base + ((unsigned)r * W + xc) * 8u);
    }
    __device__ __forceinline__ float2 resolve(const Raw& v) const { return v; }
    __device__ __forceinline__ float2 at(int r) const { return fetch(r); }
};
template <>
struct FlowSrc<true> {
    const char* base;   // coarse flow of this pair, [sh][sw][2]
    int sw, sh, sx, c1;
    float a0, a1, mul;
    double scale_y;
    struct Raw {
        float2 p00, p01, p10, p11;
        float b1;
    };
    __device__ __forceinline__ Raw fetch(int r) const
    {
        Raw v;
        float f = (float)((r + 0.5) * scale_y - 0.5);
        const int sy = floor_f(f);
        v.b1 = f - sy;
        const unsigned r0 = (unsigned)clampi(sy, 0, sh - 1) * (unsigned)sw, r1 = (unsigned)clampi(sy + 1, 0, sh - 1) * (unsigned)sw;
        v.p00 = *reinterpret_cast<const float2*>(base + (r0 + (unsigned)sx) * 8u);
        v.p01 = *reinterpret_cast<const float2*>(base + (r0 + (unsigned)c1) * 8u);
        v.p10 = *reinterpret_cast<const float2*>(base + (r1 + (unsigned)sx) * 8u);
        v.p11 = *reinterpret_cast<const float2*>(base + (r1 + (unsigned)c1) * 8u);
        return v;
    }
    __device__ __forceinline__ float2 resolve(const Raw& v) const
    {
        const float b1 = v.b1, b0 = 1.f - b1;
        float2 o;
        {
            const float t0 = v.p00.x * a0 + v.p01.x * a1, t1 = v.p10.x * a0 + v.p11.x * a1;
            o.x = (t0 * b0 + t1 * b1) * mul;
        }
        {
            const float t0 = v.p00.y * a0 + v.p01.y * a1, t1 = v.p10.y * a0 + v.p11.y * a1;
            o.y = (t0 * b0 + t1 * b1) * mul;
        }
        return o;
    }
    __device__ __forceinline__ float2 at(int r) const { return resolve(fetch(r)); }
};

// > 64 KB of dynamic LDS needs the opt-in attribute, once per (kernel instance, device); contexts of several
// devices and the worker threads of a stream pool may arrive here concurrently.
template <typename K>
int lds_opt_in(nsof_ctx* ctx, K kernel, size_t bytes)
{
    static std::atomic<unsigned long long> done{0};   // one bit per device ordinal (per template instance)
    const unsigned long long bit = 1ull << (ctx->device & 63);
    if (done.load(std::memory_order_acquire) & bit) return NSOF_OK;
    NSOF_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    done.fetch_or(bit, std::memory_order_release);
    return NSOF_OK;
}

}  // namespace
