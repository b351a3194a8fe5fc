// Farneback dense optical flow: HIP kernels for gfx950 (CDNA4, wave64).
//
// Replaces the arithmetic the reference obtains from cv2.calcOpticalFlowFarneback
// (call sites: /root/reference/optical_flow_seg.py:158,203,494 and the ob/prediction/yolo
// twins).  Every kernel keeps the reference library's operation order and float/double
// placement (see DESIGN.md "numerics contract"); this translation unit is compiled with
// -ffp-contract=off so that a*b+c stays two roundings unless fma() is written explicitly.
//
// All kernels are HBM/L2-bound stencils: no MFMA.  Layouts: images [n][h][w] f32,
// R per image: [h][w][4] f32 (channels 0-3 interleaved: one 16-B access per pixel) followed by [h][w] f32 (channel 4)
// -- the L1 serves 4 lanes per cycle whatever the access width, so the gather of the matrix update wants few, wide
// loads.  M planar [n][5][h][w] f32 (unfused path only).  Flow [n][h][w][2].
#include <cstdlib>

#include <type_traits>

#include "nsof_internal.h"

// This translation unit is compiled twice.  The regular object forms the float Gaussian blur and the bilinear resamples
// (pyramid levels, flow resize) as the library's generic C++ path does: multiply, round, add, round.  With -DNSOF_PYR_FMA
// (object farneback_kernels_fma.o) the same taps in the same order are CONTRACTED the way an AVX2+FMA3 build of the
// library's vector code (v_muladd / v_fma in its separable-filter and resize loops) contracts them: one fused
// multiply-add per tap / blend, the leading product still rounded -- the arithmetic variant twin of DESIGN.md section 2
// (context option NSOF_OPT_PYR_FMA).  Only the pyramid-level and flow-resample launchers exist in that object.
#ifdef NSOF_PYR_FMA
#define NSOF_MADD(a, b, c) fmaf((a), (b), (c))
#define NSOF_PYR_NAME(n) n##_fma
#else
#define NSOF_MADD(a, b, c) ((a) * (b) + (c))
#define NSOF_PYR_NAME(n) n
#endif

namespace {

__device__ __forceinline__ int reflect101(int p, int len)
{
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    do {
        if (p < 0) p = -p;
        else p = 2 * len - 2 - p;
    } while ((unsigned)p >= (unsigned)len);
    return p;
}
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int floor_f(float v)
{
    int i = (int)v;
    return i - (i > v);
}

// resize(INTER_LINEAR) coordinate: f = (float)((d+0.5)*scale-0.5); s = floor(f); a = f-s.
__device__ __forceinline__ void lin_coord_x(int d, double scale, int slen, int& s, float& a)
{
    float f = (float)((d + 0.5) * scale - 0.5);
    s = floor_f(f);
    a = f - s;
    if (s < 0) { s = 0; a = 0.f; }
    if (s >= slen - 1) { s = slen - 1; a = 0.f; }
}
__device__ __forceinline__ void lin_coord_y(int d, double scale, int& s, float& a)
{
    float f = (float)((d + 0.5) * scale - 0.5);
    s = floor_f(f);
    a = f - s;
}

// ---------------------------------------------------------------------------------------
// Pyramid level preparation: u8 -> f32, separable Gaussian (sepFilter2D order), bilinear
// resample of the blurred FULL-RES image to (wk, hk).
// ---------------------------------------------------------------------------------------

// Row filter at an (unreflected) column c of one row, ordering per kernel size.
// KS > 0: kernel size known at compile time (loops unroll, taps are scalar registers); KS == 0: runtime size,
// the tap accessor `tk(j)` then reads a copy of the taps in LDS (a dynamically indexed kernel argument, or a
// pointer to it, would be one dependent memory load per tap -- measured 10x slower).
template <int KS, typename TapF, typename LoadF>
__device__ __forceinline__ float row_filter(TapF tk, int ksize, int c, LoadF ld)
{
    const int ks = KS ? KS : ksize, r = ks >> 1;
    if (ks == 3) return NSOF_MADD(ld(c - 1) + ld(c + 1), tk(2), ld(c) * tk(1));
    if (ks == 5) return NSOF_MADD(ld(c - 2) + ld(c + 2), tk(4), NSOF_MADD(ld(c - 1) + ld(c + 1), tk(3), ld(c) * tk(2)));
    float s = tk(0) * ld(c - r);
#pragma unroll
    for (int j = 1; j < ks; j++) s = NSOF_MADD(tk(j), ld(c - r + j), s);
    return s;
}
// Column filter at (unreflected) row rr given an accessor of row-filtered values.
template <int KS, typename TapF, typename LoadF>
__device__ __forceinline__ float col_filter(TapF tk, int ksize, int rr, LoadF hv)
{
    const int ks = KS ? KS : ksize, r = ks >> 1;
    if (ks == 3) return NSOF_MADD(hv(rr - 1) + hv(rr + 1), tk(2), hv(rr) * tk(1));
    float s = tk(r) * hv(rr);
#pragma unroll
    for (int j = 1; j <= r; j++) s = NSOF_MADD(tk(r + j), hv(rr + j) + hv(rr - j), s);
    return s;
}

// Geometry of one image of a pyramid-level launch.  HET == false: the uniform batch (every image has the kernel
// arguments' shape, image z lives at src + z*img_stride and goes to out + z*wk*hk).  HET == true: image z belongs to
// work item z/2 of the device table (z & 1: prev / next), see nsof_het_item.
struct PrepImg {
    const uint8_t* img;
    float* dst;
};
template <bool HET>
__device__ __forceinline__ bool prep_geom(PrepImg& g, const uint8_t* src, ptrdiff_t& row_stride, ptrdiff_t img_stride,
                                          int& W, int& H, int& wk, int& hk, float* out,
                                          const nsof_het_item* __restrict__ items, int want_flag)
{
    if constexpr (HET) {
        const nsof_het_item& it = items[blockIdx.z >> 1];
        const int which = blockIdx.z & 1;
        if (want_flag >= 0 && (it.flags & NSOF_HET_VEC0) != want_flag) return false;
        W = it.W; H = it.H; wk = it.wk; hk = it.hk;
        row_stride = (ptrdiff_t)it.src_stride[which];
        g.img = it.src[which];
        g.dst = out + it.offI + (size_t)which * wk * hk;
    } else {
        g.img = src + (ptrdiff_t)blockIdx.z * img_stride;
        g.dst = out + (size_t)blockIdx.z * wk * hk;
    }
    return true;
}

// Same-size level (k = 0), generic: one thread per pixel, no resample.
template <bool HET>
__global__ __launch_bounds__(256) void k_prep_same(const uint8_t* __restrict__ src, ptrdiff_t row_stride,
                                                    ptrdiff_t img_stride, int W, int H, nsof_blur_taps t,
                                                    float* __restrict__ out, const nsof_het_item* __restrict__ items,
                                                    int want_flag)
{
    __shared__ float s_tk[NSOF_MAX_BLUR_TAPS];
    if (threadIdx.x < NSOF_MAX_BLUR_TAPS) s_tk[threadIdx.x] = t.k[threadIdx.x];
    __syncthreads();
    auto tk = [&](int j) { return s_tk[j]; };
    PrepImg g;
    int wk = W, hk = H;
    if (!prep_geom<HET>(g, src, row_stride, img_stride, W, H, wk, hk, out, items, want_flag)) return;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const uint8_t* img = g.img;
    auto hv = [&](int rr) {
        const uint8_t* rowp = img + (ptrdiff_t)reflect101(rr, H) * row_stride;
        return row_filter<0>(tk, t.ksize, x, [&](int c) { return (float)rowp[reflect101(c, W)]; });
    };
    g.dst[(size_t)y * W + x] = col_filter<0>(tk, t.ksize, y, hv);
}

// Same-size level with the 3-tap kernel (every level 0): a lane owns 4 adjacent pixels of 8 consecutive rows.
// One aligned dword load per row; the two bytes outside the dword come from the neighbouring lanes
// (the wave's edge lanes fetch theirs from memory); row-filter results are shared between the three
// output rows that use them; 16-B stores.  Requires 4-byte aligned rows (else k_prep_same).
constexpr int PREP0_ROWS = 8;
template <bool HET>
__global__ __launch_bounds__(256) void k_prep_same3_vec(const uint8_t* __restrict__ src, ptrdiff_t row_stride,
                                                         ptrdiff_t img_stride, int W, int H, float k0, float k1,
                                                         float* __restrict__ out,
                                                         const nsof_het_item* __restrict__ items)
{
    PrepImg g;
    int wk = W, hk = H;
    if (!prep_geom<HET>(g, src, row_stride, img_stride, W, H, wk, hk, out, items, NSOF_HET_VEC0)) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = (blockIdx.x * 64 + lane) * 4;                  // first of this lane's 4 pixels
    const int y0 = (blockIdx.y * 4 + wave) * PREP0_ROWS;
    if (y0 >= H || blockIdx.x * 256 >= W) return;                 // wave-uniform
    const bool live = x < W;
    const int xl = live ? x : 0;                                  // dead lanes still take part in the shuffles
    const uint8_t* img = g.img;
    float* dst = g.dst;

    auto hrow = [&](int rr, float (&h)[4]) {                      // row filter of source row rr (reflected)
        const uint8_t* rowp = img + (ptrdiff_t)reflect101(rr, H) * row_stride;
        const unsigned v = *reinterpret_cast<const unsigned*>(rowp + xl);
        unsigned lft = __shfl_up(v, 1) >> 24, rgt = __shfl_down(v, 1) & 0xffu;
        if (lane == 0 || x == 0) lft = rowp[reflect101(xl - 1, W)];
        if (lane == 63 || x + 4 >= W) rgt = rowp[reflect101(xl + 4, W)];
        const float s0 = (float)(v & 0xffu), s1 = (float)((v >> 8) & 0xffu), s2 = (float)((v >> 16) & 0xffu),
                    s3 = (float)(v >> 24), sl = (float)lft, sr = (float)rgt;
        h[0] = NSOF_MADD(sl + s1, k1, s0 * k0);
        h[1] = NSOF_MADD(s0 + s2, k1, s1 * k0);
        h[2] = NSOF_MADD(s1 + s3, k1, s2 * k0);
        h[3] = NSOF_MADD(s2 + sr, k1, s3 * k0);
    };
    float hm[4], h0[4], hp[4];
    hrow(y0 - 1, hm);
    hrow(y0, h0);
#pragma unroll
    for (int q = 0; q < PREP0_ROWS; q++) {
        const int y = y0 + q;
        if (y >= H) break;                                        // wave-uniform
        hrow(y + 1, hp);
        if (live) {
            float4 o;
            o.x = NSOF_MADD(hm[0] + hp[0], k1, h0[0] * k0);
            o.y = NSOF_MADD(hm[1] + hp[1], k1, h0[1] * k0);
            o.z = NSOF_MADD(hm[2] + hp[2], k1, h0[2] * k0);
            o.w = NSOF_MADD(hm[3] + hp[3], k1, h0[3] * k0);
            nsof_store_stream4(dst + (size_t)y * W + x, o.x, o.y, o.z, o.w);
        }
#pragma unroll
        for (int i = 0; i < 4; i++) { hm[i] = h0[i]; h0[i] = hp[i]; }
    }
}

// Resampled level for EXACT decimation by S = 2, 4, 8 (W = S*wk, H = S*hk, W % 16 == 0: the pyr_scale 0.5 pyramids
// of 1080p/4K frames).  The thread-per-output kernels below issue dozens of narrow loads per output pixel and are
// bound by the L1's 4 lanes/cycle; here a lane owns 16 adjacent source columns (16/S output pixels) and walks down
// the output rows of its segment:
//   * one aligned 16-B load per source row + one halo load per side (the image's own edge reflects out of the
//     lane's 16 bytes, no extra load),
//   * the row filter is evaluated only at the columns the resize samples (S*j + S/2 - 1 and the next one), once
//     per source row, and kept in a register ring of RING >= KS + 1 rows (static slots: the walk is unrolled
//     over RING / S steps),
//   * column filter at the two sampled rows, then resize's 2x2 blend (all four weights are exactly 0.5).
// Same operation order as the other prep kernels (row_filter / col_filter), so the output is bit-identical.
// The walk of one wave: column group `bx * 64 + lane`, output rows [seg * seg_rows, (seg + 1) * seg_rows) of image z.
template <int S, int KS, int CW>
__device__ __forceinline__ void prep_decim_body(const uint8_t* __restrict__ src, ptrdiff_t row_stride, ptrdiff_t img_stride,
                                                int W, int H, int wk, int hk, int seg_rows, const nsof_blur_taps& t,
                                                float* __restrict__ out, int bx, int seg, int z)
{
    // CW = source columns per lane: 16 (one 16-B load per row) when W % 16 == 0, else 8 (W % 8 == 0, e.g. the
    // 1080-wide portrait frames of the reference's grasp sequence)
    constexpr int R = KS / 2, NPX = CW / S, NC = 2 * NPX;
    constexpr int RING = (KS + 1 + S - 1) / S * S, U = RING / S;
    constexpr int HB = (R + 3) / 4 * 4, HD = HB / 4;          // halo bytes / dwords per side
    constexpr int WIN = HB + CW + HB;
    static_assert(NPX >= 1, "lane narrower than one output pixel");
    const int lane = threadIdx.x & 63;
    const int T = bx * 64 + lane;                              // CW-column group
    const int dy0 = seg * seg_rows;
    if (dy0 >= hk) return;                                     // wave-uniform
    const int dy1 = min(dy0 + seg_rows, hk);
    const bool live = CW * T < W;
    const int xc0 = live ? CW * T : 0;
    const bool edge_l = xc0 == 0, edge_r = xc0 + CW >= W;
    const uint8_t* img = src + (ptrdiff_t)z * img_stride;
    float* dst = out + (size_t)z * wk * hk;
    auto tk = [&](int j) { return t.k[j]; };                   // static index after unrolling

    float ring[RING][NC];

    // A source row as loaded: the lane's CW bytes + HB halo bytes per side.  fetch() only issues the loads, filt()
    // evaluates the row filter at this lane's sampled columns -> ring[slot]; between the two a row can stay in flight
    // while the rows before it are filtered (S <= 4: the rows of the NEXT output row are fetched before this one's are
    // used -- with one output row's 2 or 4 source rows in flight per wave the kernel waited for memory half the time).
    struct Raw {
        unsigned cw[CW / 4], hl[HD], hr[HD];
    };
    auto fetch = [&](int r, Raw& q) {
        const uint8_t* rowp = img + (ptrdiff_t)reflect101(r, H) * row_stride + xc0;
        if (CW == 16) {
            const uint4 c = *reinterpret_cast<const uint4*>(rowp);
            q.cw[0] = c.x; q.cw[1] = c.y; q.cw[(CW / 4 > 2) ? 2 : 0] = c.z; q.cw[(CW / 4 > 3) ? 3 : 0] = c.w;
        } else {
            const uint2 c = *reinterpret_cast<const uint2*>(rowp);
            q.cw[0] = c.x; q.cw[1] = c.y;
        }
        const uint8_t* lp = edge_l ? rowp : rowp - HB;          // edge lanes: any valid address, value unused
        const uint8_t* rp = edge_r ? rowp : rowp + CW;
#pragma unroll
        for (int d = 0; d < HD; d++) {
            q.hl[d] = reinterpret_cast<const unsigned*>(lp)[d];
            q.hr[d] = reinterpret_cast<const unsigned*>(rp)[d];
        }
    };
    auto filt = [&](const Raw& q, float (&dstrow)[NC]) {
        float raw[WIN];   // window [xc0 - HB, xc0 + CW + HB) as loaded
#pragma unroll
        for (int b = 0; b < HB; b++) {
            raw[b] = (float)((q.hl[b >> 2] >> (8 * (b & 3))) & 0xffu);
            raw[HB + CW + b] = (float)((q.hr[b >> 2] >> (8 * (b & 3))) & 0xffu);
        }
#pragma unroll
        for (int b = 0; b < CW; b++) raw[HB + b] = (float)((q.cw[b >> 2] >> (8 * (b & 3))) & 0xffu);
        float fb[WIN];
#pragma unroll
        for (int b = 0; b < CW; b++) fb[HB + b] = raw[HB + b];
#pragma unroll
        for (int b = 0; b < HB; b++) {
            // left halo byte b is column xc0 - HB + b; at the image edge (xc0 == 0) it reflects to column HB - b
            fb[b] = edge_l ? raw[HB + (HB - b)] : raw[b];
            // right halo byte b is column xc0 + CW + b; at the edge (xc0 + CW == W) it reflects to column W - 2 - b
            fb[HB + CW + b] = edge_r ? raw[HB + CW - 2 - b] : raw[HB + CW + b];
        }
#pragma unroll
        for (int n = 0; n < NC; n++) {
            const int off = HB + S * (n >> 1) + S / 2 - 1 + (n & 1);
            dstrow[n] = row_filter<KS>(tk, KS, off, [&](int q2) { return fb[q2]; });
        }
    };
    auto load_row = [&](int r, float (&dstrow)[NC]) {
        Raw q;
        fetch(r, q);
        filt(q, dstrow);
    };
#ifndef NSOF_DECIM_PF
#define NSOF_DECIM_PF 4
#endif
    constexpr bool PF = S <= NSOF_DECIM_PF;   // prefetch one output row ahead

    // relative row index rel = r - base, base = first row needed by output row dy0; slot = rel % RING
    const int base = S * dy0 + S / 2 - 1 - R;
#pragma unroll
    for (int i = 0; i <= KS - S; i++) load_row(base + i, ring[i % RING]);

    Raw nx[PF ? S : 1];
    if constexpr (PF) {
#pragma unroll
        for (int i = 0; i < S; i++) fetch(base + KS - S + 1 + i, nx[i]);
    }
    for (int g = 0; dy0 + g * U < dy1; g++) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int dy = dy0 + g * U + u;
            if (dy >= dy1) break;                               // wave-uniform
            if constexpr (PF) {
                Raw cu[S];
#pragma unroll
                for (int i = 0; i < S; i++) cu[i] = nx[i];
                // the next output row's source rows (beyond the segment: reflected rows of the image, never used)
#pragma unroll
                for (int i = 0; i < S; i++) fetch(base + S * (g * U + u + 1) + KS - S + 1 + i, nx[i]);
#pragma unroll
                for (int i = 0; i < S; i++) filt(cu[i], ring[(S * u + KS - S + 1 + i) % RING]);
            } else {
#pragma unroll
                for (int i = 0; i < S; i++) {
                    load_row(base + S * (g * U + u) + KS - S + 1 + i, ring[(S * u + KS - S + 1 + i) % RING]);
                }
            }
            if (live) {
                float o[NPX];
#pragma unroll
                for (int j = 0; j < NPX; j++) {
                    auto col = [&](int n, int centre) {
                        return col_filter<KS>(tk, KS, centre, [&](int q) { return ring[(S * u + q) % RING][n]; });
                    };
                    const float B00 = col(2 * j, R), B01 = col(2 * j + 1, R);
                    const float B10 = col(2 * j, R + 1), B11 = col(2 * j + 1, R + 1);
                    const float t0 = B00 * 0.5f + B01 * 0.5f;
                    const float t1 = B10 * 0.5f + B11 * 0.5f;
                    o[j] = t0 * 0.5f + t1 * 0.5f;
                }
                float* op = dst + (size_t)dy * wk + NPX * T;
                if (NPX == 8) {
                    nsof_store_stream4(op, o[0], o[1 % NPX], o[2 % NPX], o[3 % NPX]);
                    nsof_store_stream4(op + 4, o[4 % NPX], o[5 % NPX], o[6 % NPX], o[7 % NPX]);
                } else if (NPX == 4) {
                    nsof_store_stream4(op, o[0], o[1 % NPX], o[2 % NPX], o[3 % NPX]);
                } else if (NPX == 2) {
                    nsof_store_stream2(op, o[0], o[1 % NPX]);
                } else {
                    __builtin_nontemporal_store(o[0], op);
                }
            }
        }
    }
}

template <int S, int KS, int CW>
__global__ __launch_bounds__(256) void k_prep_decim(const uint8_t* __restrict__ src, ptrdiff_t row_stride,
                                                     ptrdiff_t img_stride, int W, int H, int wk, int hk, int seg_rows,
                                                     nsof_blur_taps t, float* __restrict__ out)
{
    prep_decim_body<S, KS, CW>(src, row_stride, img_stride, W, H, wk, hk, seg_rows, t, out, blockIdx.x,
                               blockIdx.y * 4 + (threadIdx.x >> 6), blockIdx.z);
}

// Levels 1, 2, 3 of a pyr_scale 0.5 pyramid in ONE launch: twelve waves per workgroup, four per level, all over the same
// 1024 (512) source columns and the same 4 x 96 source rows -- each level smooths the FULL-RES frame, so run as three
// launches the frame crossed the fabric three times; side by side the second and third reader find its rows in the cache.
// Same walks, same bits (prep_decim_body).
struct Decim3 {
    nsof_blur_taps t[3];
    float* out[3];
    int seg_rows[3];
};
template <int CW>
__global__ __launch_bounds__(768) void k_prep_decim3(const uint8_t* __restrict__ src, ptrdiff_t row_stride,
                                                      ptrdiff_t img_stride, int W, int H, Decim3 d)
{
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int seg = blockIdx.y * 4 + (wave & 3);
    if (wave < 4)
        prep_decim_body<2, 3, CW>(src, row_stride, img_stride, W, H, W / 2, H / 2, d.seg_rows[0], d.t[0], d.out[0], blockIdx.x, seg, blockIdx.z);
    else if (wave < 8)
        prep_decim_body<4, 9, CW>(src, row_stride, img_stride, W, H, W / 4, H / 4, d.seg_rows[1], d.t[1], d.out[1], blockIdx.x, seg, blockIdx.z);
    else
        prep_decim_body<8, 19, CW>(src, row_stride, img_stride, W, H, W / 8, H / 8, d.seg_rows[2], d.t[2], d.out[2], blockIdx.x, seg, blockIdx.z);
}

// Resampled level, generic fallback: one thread per destination pixel, no data sharing.
template <bool HET>
__global__ __launch_bounds__(256) void k_prep_naive(const uint8_t* __restrict__ src, ptrdiff_t row_stride,
                                                     ptrdiff_t img_stride, int W, int H, int wk, int hk,
                                                     double scale_x, double scale_y, nsof_blur_taps t,
                                                     float* __restrict__ out, const nsof_het_item* __restrict__ items)
{
    __shared__ float s_tk[NSOF_MAX_BLUR_TAPS];
    if (threadIdx.x < NSOF_MAX_BLUR_TAPS) s_tk[threadIdx.x] = t.k[threadIdx.x];
    __syncthreads();
    auto tk = [&](int j) { return s_tk[j]; };
    PrepImg g;
    prep_geom<HET>(g, src, row_stride, img_stride, W, H, wk, hk, out, items, -1);
    if constexpr (HET) { scale_x = 1. / ((double)wk / W); scale_y = 1. / ((double)hk / H); }
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dx >= wk || dy >= hk) return;
    const uint8_t* img = g.img;
    int sx, sy;
    float a1, b1;
    lin_coord_x(dx, scale_x, W, sx, a1);
    lin_coord_y(dy, scale_y, sy, b1);
    const float a0 = 1.f - a1, b0 = 1.f - b1;
    const int c0 = sx, c1 = min(sx + 1, W - 1);
    const int r0 = clampi(sy, 0, H - 1), r1 = clampi(sy + 1, 0, H - 1);
    auto blur = [&](int rr, int cc) {
        auto hv = [&](int q) {
            const uint8_t* rowp = img + (ptrdiff_t)reflect101(q, H) * row_stride;
            return row_filter<0>(tk, t.ksize, cc, [&](int c) { return (float)rowp[reflect101(c, W)]; });
        };
        return col_filter<0>(tk, t.ksize, rr, hv);
    };
    const float t0 = NSOF_MADD(blur(r0, c0), a0, blur(r0, c1) * a1);
    const float t1 = NSOF_MADD(blur(r1, c0), a0, blur(r1, c1) * a1);
    g.dst[(size_t)dy * wk + dx] = NSOF_MADD(t0, b0, t1 * b1);
}

// Resampled level, LDS-tiled: a 32x8 destination tile per 256-thread block.
//   phase 1: source footprint (with blur halo, borders reflected) -> LDS as u8 (coalesced row segments)
//   phase 2: row filter only at the 2 source columns each destination column samples
//   phase 3: column filter only at the 2 source rows each destination row samples
//   phase 4: bilinear blend (horizontal first, then vertical, as resize does)
// KS = compile-time kernel size (3, 5, 9, 19: pyr_scale 0.5 / 0.6 with up to 3 levels) or 0 = runtime.
constexpr int PREP_TW = 32, PREP_TH = 8;
template <int KS, bool HET>
__global__ __launch_bounds__(256) void k_prep_tiled(const uint8_t* __restrict__ src, ptrdiff_t row_stride,
                                                     ptrdiff_t img_stride, int W, int H, int wk, int hk,
                                                     double scale_x, double scale_y, int rw_cap, int rh_cap,
                                                     nsof_blur_taps t, float* __restrict__ out,
                                                     const nsof_het_item* __restrict__ items)
{
    PrepImg g;
    prep_geom<HET>(g, src, row_stride, img_stride, W, H, wk, hk, out, items, -1);
    if constexpr (HET) {
        scale_x = 1. / ((double)wk / W);
        scale_y = 1. / ((double)hk / H);
        if (blockIdx.x * PREP_TW >= wk || blockIdx.y * PREP_TH >= hk) return;   // block-uniform
    }
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* sH = reinterpret_cast<float*>(smem);                 // [rh_cap][2*TW]
    float* sB = sH + (size_t)rh_cap * (2 * PREP_TW);            // [2*TH][2*TW]
    unsigned char* sU = reinterpret_cast<unsigned char*>(sB + 2 * PREP_TH * 2 * PREP_TW);  // [rh_cap][rw_cap]
    __shared__ int s_c[2 * PREP_TW];   // absolute source column per (dst col, 0/1)
    __shared__ int s_r[2 * PREP_TH];   // absolute source row per (dst row, 0/1)
    __shared__ float s_a[PREP_TW], s_b[PREP_TH];
    __shared__ float s_tk[NSOF_MAX_BLUR_TAPS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ksize = KS ? KS : t.ksize, r = ksize >> 1;
    const int dx0 = blockIdx.x * PREP_TW, dy0 = blockIdx.y * PREP_TH;
    const uint8_t* img = g.img;
    auto tk = [&](int j) { return KS ? t.k[j] : s_tk[j]; };   // KS > 0: j is a constant after unrolling

    if (tid < PREP_TW) {
        int dx = min(dx0 + tid, wk - 1), sx;
        float a;
        lin_coord_x(dx, scale_x, W, sx, a);
        s_c[2 * tid] = sx;
        s_c[2 * tid + 1] = min(sx + 1, W - 1);
        s_a[tid] = a;
    } else if (tid >= 64 && tid < 64 + PREP_TH) {
        int i = tid - 64, dy = min(dy0 + i, hk - 1), sy;
        float b;
        lin_coord_y(dy, scale_y, sy, b);
        s_r[2 * i] = clampi(sy, 0, H - 1);
        s_r[2 * i + 1] = clampi(sy + 1, 0, H - 1);
        s_b[i] = b;
    } else if (!KS && tid >= 128 && tid < 128 + NSOF_MAX_BLUR_TAPS) {
        s_tk[tid - 128] = t.k[tid - 128];
    }
    __syncthreads();
    // coordinates are monotone in dx/dy, so the footprint is [first .. last]
    int C0 = s_c[0] - r;
    const int RW0 = s_c[2 * PREP_TW - 1] + r - C0 + 1;
    const int R0 = s_r[0] - r, RH = s_r[2 * PREP_TH - 1] + r - R0 + 1;
    // host sized rw_cap/rh_cap from the same arithmetic (+4 columns of slack for the aligned copy below)
    const bool interior = C0 >= 0 && C0 + RW0 <= W && R0 >= 0 && R0 + RH <= H && (W & 3) == 0 &&
                          (row_stride & 3) == 0 && (reinterpret_cast<uintptr_t>(img) & 3) == 0;
    if (interior) {
        // no border inside the footprint: copy whole aligned dwords (64 lanes x 4 B per wave-instruction)
        const int C0a = C0 & ~3, nd = (C0 + RW0 - C0a + 3) >> 2;
        C0 = C0a;   // the LDS image now starts at the aligned column
#pragma unroll 4
        for (int rr = wave; rr < RH; rr += 4) {
            const uint8_t* rowp = img + (ptrdiff_t)(R0 + rr) * row_stride + C0a;
            for (int d = lane; d < nd; d += 64)
                *reinterpret_cast<unsigned*>(sU + rr * rw_cap + 4 * d) = *reinterpret_cast<const unsigned*>(rowp + 4 * d);
        }
    } else {
        for (int rr = wave; rr < RH; rr += 4) {          // border tile: per byte, BORDER_REFLECT_101
            const uint8_t* rowp = img + (ptrdiff_t)reflect101(R0 + rr, H) * row_stride;
            for (int cc = lane; cc < RW0; cc += 64) sU[rr * rw_cap + cc] = rowp[reflect101(C0 + cc, W)];
        }
    }
    __syncthreads();
    {
        const int cj = s_c[lane] - C0;                // this lane's sampled column, local
#pragma unroll 2
        for (int rr = wave; rr < RH; rr += 4) {
            const unsigned char* rowp = sU + rr * rw_cap;
            sH[rr * (2 * PREP_TW) + lane] = row_filter<KS>(tk, ksize, cj, [&](int c) { return (float)rowp[c]; });
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = wave; q < 2 * PREP_TH; q += 4)       // 16 sampled rows x 64 sampled columns
        sB[q * (2 * PREP_TW) + lane] =
            col_filter<KS>(tk, ksize, s_r[q] - R0, [&](int rr) { return sH[rr * (2 * PREP_TW) + lane]; });
    __syncthreads();
    const int tx = tid & 31, ty = tid >> 5;
    const int dx = dx0 + tx, dy = dy0 + ty;
    if (dx < wk && dy < hk) {
        const float a1 = s_a[tx], a0 = 1.f - a1, b1 = s_b[ty], b0 = 1.f - b1;
        const float* B0 = sB + (2 * ty) * (2 * PREP_TW) + 2 * tx;
        const float* B1 = B0 + 2 * PREP_TW;
        const float t0 = NSOF_MADD(B0[0], a0, B0[1] * a1);
        const float t1 = NSOF_MADD(B1[0], a0, B1[1] * a1);
        g.dst[(size_t)dy * wk + dx] = NSOF_MADD(t0, b0, t1 * b1);
    }
}

// Resampled level, direct: one thread per destination pixel, everything in registers, no LDS, no barriers.
// A destination pixel blends the blurred image at 2x2 source positions (rows r0,r1 x columns c0,c1), i.e. it
// needs the row-filtered values H at columns c0 and c1 of the KS+1 source rows r0-R..r1+R; each of those rows
// contributes KS+1 consecutive bytes, fetched as unaligned dwords (L1/L2 resident: the u8 frame is 2 MB).
// More arithmetic than the LDS-tiled variant but no per-tile overhead -- measured 3-6x faster at 1080p.
// Pixels whose footprint leaves the image take a per-byte path with BORDER_REFLECT_101 indexing.
template <int KS, bool HET>
__global__ __launch_bounds__(256) void k_prep_direct(const uint8_t* __restrict__ src, ptrdiff_t row_stride,
                                                      ptrdiff_t img_stride, int W, int H, int wk, int hk,
                                                      double scale_x, double scale_y, nsof_blur_taps t,
                                                      float* __restrict__ out, const nsof_het_item* __restrict__ items)
{
    constexpr int R = KS / 2, NB = KS + 1, ND = (NB + 3) / 4;
    PrepImg g;
    prep_geom<HET>(g, src, row_stride, img_stride, W, H, wk, hk, out, items, -1);
    if constexpr (HET) { scale_x = 1. / ((double)wk / W); scale_y = 1. / ((double)hk / H); }
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dx >= wk || dy >= hk) return;
    const uint8_t* img = g.img;
    int sx, sy;
    float a1, b1;
    lin_coord_x(dx, scale_x, W, sx, a1);
    lin_coord_y(dy, scale_y, sy, b1);
    const float a0 = 1.f - a1, b0 = 1.f - b1;
    const int c0 = sx, c1 = min(sx + 1, W - 1);
    const int r0 = clampi(sy, 0, H - 1), r1 = clampi(sy + 1, 0, H - 1);
    const bool fast = c0 - R >= 0 && c0 - R + 4 * ND <= W && c1 == c0 + 1;
    auto tk = [&](int j) { return t.k[j]; };   // j is a compile-time constant after unrolling

    float H0[KS + 1], H1[KS + 1];
#pragma unroll
    for (int i = 0; i <= KS; i++) {
        const uint8_t* rowp = img + (ptrdiff_t)reflect101(r0 - R + i, H) * row_stride;
        float b[NB], bb[NB];   // bytes around c0 and around c1
        if (fast) {
#pragma unroll
            for (int d = 0; d < ND; d++) {
                unsigned v;
                __builtin_memcpy(&v, rowp + (c0 - R) + 4 * d, 4);   // unaligned dword load
#pragma unroll
                for (int e = 0; e < 4; e++)
                    if (4 * d + e < NB) b[4 * d + e] = (float)((v >> (8 * e)) & 0xffu);
            }
            H0[i] = row_filter<KS>(tk, KS, R, [&](int c) { return b[c]; });
            H1[i] = row_filter<KS>(tk, KS, R + 1, [&](int c) { return b[c]; });
        } else {
#pragma unroll
            for (int j = 0; j < KS; j++) {
                b[j] = (float)rowp[reflect101(c0 - R + j, W)];
                bb[j] = (float)rowp[reflect101(c1 - R + j, W)];
            }
            H0[i] = row_filter<KS>(tk, KS, R, [&](int c) { return b[c]; });
            H1[i] = row_filter<KS>(tk, KS, R, [&](int c) { return bb[c]; });
        }
    }
    // rows of H0/H1 are r0-R .. r0-R+KS; the window of r1 = r0+1 starts one entry later
    const float B00 = col_filter<KS>(tk, KS, R, [&](int q) { return H0[q]; });
    const float B01 = col_filter<KS>(tk, KS, R, [&](int q) { return H1[q]; });
    float B10 = B00, B11 = B01;
    if (r1 != r0) {
        B10 = col_filter<KS>(tk, KS, R + 1, [&](int q) { return H0[q]; });
        B11 = col_filter<KS>(tk, KS, R + 1, [&](int q) { return H1[q]; });
    }
    const float t0 = NSOF_MADD(B00, a0, B01 * a1);
    const float t1 = NSOF_MADD(B10, a0, B11 * a1);
    g.dst[(size_t)dy * wk + dx] = NSOF_MADD(t0, b0, t1 * b1);
}

// Resampled level, walking: thread <-> destination column, a wave walks a segment of destination rows top to bottom.
// The direct kernel above recomputes, for EVERY destination pixel, the row filter of its KS+1 source rows (one or more
// unaligned dword loads each); consecutive destination rows of a column share most of those rows.  Here a thread keeps the
// row-filtered values of its two sampled columns for a window of KS+1 source rows in registers (H0 / H1, shifted as the
// window advances -- by 1 or 2 rows per destination row at pyr_scale 0.6, 4-5 at level 3) and loads / row-filters every
// source row ONCE: 1.7 instead of 4 row evaluations per destination pixel at level 1, 4.6 instead of 10 at level 3.  The
// advance loop is wave-uniform (source rows depend on the destination row only).  Same helper functions and operation
// order as the direct kernel -> bit-identical output.  Parameter sets B / C (pyr_scale 0.6): levels 1 and 2 (3 / 5 taps).
// Round 4: the walk is written for the scalar unit.  Everything that depends on the destination ROW only -- the source
// rows of its window, "has the window's last row entered", the vertical blend weight -- is wave-uniform and now lives in
// SGPRs (readfirstlane), so the loop's control flow is scalar branches instead of exec-mask bookkeeping; and whether a
// lane may take the unaligned-dword fast path (its KS + 1 bytes lie inside the image) is decided once per WAVE: only the
// waves that touch the left / right image border run the per-byte reflecting loads.  The first version spent ~240
// instructions per source row, most of them mask handling around the per-lane fast / slow choice (ISA: 54 s_cbranch_execz,
// 53 s_and_saveexec, 47 v_cndmask, 24 global_load_ubyte per 4 source rows), where the arithmetic needs ~20.
template <int U, int N, class F>
__device__ __forceinline__ void prep_static_slots(F& f)
{
    if constexpr (U < N) {
        if (f(std::integral_constant<int, U>{})) prep_static_slots<U + 1, N>(f);
    }
}

template <int KS, bool WFAST>
__device__ __forceinline__ void prep_walk_body(const uint8_t* __restrict__ img, ptrdiff_t row_stride, int W, int H, int wk, int hk,
                                               double scale_y, int dy0, int dy_end, int dx, bool live, int c0, int c1, bool fast,
                                               float a1, const nsof_blur_taps& t, float* __restrict__ dst)
{
    constexpr int R = KS / 2, NB = KS + 1, ND = (NB + 3) / 4;
    const float a0 = 1.f - a1;
    auto tk = [&](int j) { return t.k[j]; };   // j is a compile-time constant after unrolling
    const unsigned coff = (unsigned)(c0 - R);    // byte offset of this lane's window in a source row (fast lanes)
    // the KS + 1 bytes of source row rr (wave-uniform, any integer: reflected) around this lane's two columns, as raw dwords
    auto load_row = [&](int rr, unsigned (&raw)[ND]) {
        const uint8_t* rowp = img + (ptrdiff_t)reflect101(rr, H) * row_stride;   // scalar
#pragma unroll
        for (int d = 0; d < ND; d++) __builtin_memcpy(&raw[d], rowp + coff + 4 * d, 4);   // unaligned dword loads
    };
    // row-filtered values at columns c0 and c1 from those bytes
    auto filt_row = [&](const unsigned (&raw)[ND], float& h0, float& h1) {
        float b[NB];
#pragma unroll
        for (int d = 0; d < ND; d++)
#pragma unroll
            for (int e = 0; e < 4; e++)
                if (4 * d + e < NB) b[4 * d + e] = (float)((raw[d] >> (8 * e)) & 0xffu);
        h0 = row_filter<KS>(tk, KS, R, [&](int c) { return b[c]; });
        h1 = row_filter<KS>(tk, KS, R + 1, [&](int c) { return b[c]; });
    };
    // border waves: per lane, per byte with BORDER_REFLECT_101 where the window leaves the image
    auto hrow_edge = [&](int rr, float& h0, float& h1) {
        const uint8_t* rowp = img + (ptrdiff_t)reflect101(rr, H) * row_stride;
        if (fast) {
            unsigned raw[ND];
#pragma unroll
            for (int d = 0; d < ND; d++) __builtin_memcpy(&raw[d], rowp + coff + 4 * d, 4);
            filt_row(raw, h0, h1);
        } else {
            float b[NB], bb[NB];
#pragma unroll
            for (int j = 0; j < KS; j++) {
                b[j] = (float)rowp[reflect101(c0 - R + j, W)];
                bb[j] = (float)rowp[reflect101(c1 - R + j, W)];
            }
            h0 = row_filter<KS>(tk, KS, R, [&](int c) { return b[c]; });
            h1 = row_filter<KS>(tk, KS, R, [&](int c) { return bb[c]; });
        }
    };
    // destination row dy -> its two source rows and the vertical weight, all wave-uniform (SGPRs)
    auto row_coord = [&](int dy, int& r0, int& r1, float& b1) {
        int sy;
        float bv;
        lin_coord_y(dy, scale_y, sy, bv);
        sy = __builtin_amdgcn_readfirstlane(sy);
        b1 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, bv)));
        r0 = clampi(sy, 0, H - 1);
        r1 = clampi(sy + 1, 0, H - 1);
    };
    // The row-filtered pairs of the KS + 1 newest source rows live in a register ring with STATIC slots: source rows are
    // taken strictly in order, row rstart + i into slot i % NB (the walk is unrolled NB times), and a destination row is
    // emitted when the last row of its window, r0 + R + 1, has just entered -- its window rows then sit at the slots
    // (u + 1 + q) % NB, q = 0..KS, known at compile time.  (Windows of consecutive destination rows overlap -- the launcher
    // takes this kernel only while a destination step is shorter than the ring -- so no source row is filtered in vain.)
    float H0[NB], H1[NB];
    int dy = dy0, r0, r1;
    float b1;
    row_coord(dy, r0, r1, b1);
    const int rstart = r0 - R;
    int last = r0 - R + KS;                                        // the source row that completes the window of dy
    // (interior waves) the raw bytes of the NEXT block of KS + 1 source rows are requested before this block is filtered:
    // a wave then has a block of loads in flight instead of waiting for every row's load on its own
    unsigned cur[NB][ND], nxt[NB][ND];
    if constexpr (WFAST) {
#pragma unroll
        for (int u = 0; u < NB; u++) load_row(rstart + u, cur[u]);
    }
    for (int rb = 0; dy < dy_end && rb <= H + 2 * NB; rb += NB) {   // (bounded: at most H + KS + 1 source rows are walked)
        if constexpr (WFAST) {
#pragma unroll
            for (int u = 0; u < NB; u++) load_row(rstart + rb + NB + u, nxt[u]);   // rows beyond the segment: reflected, unused
        }
        // (expanded at compile time: with 10 slots a `#pragma unroll` loop stayed rolled and indexed the ring dynamically)
        auto slot = [&](auto uc) -> bool {
            constexpr int u = decltype(uc)::value;
            if (dy >= dy_end) return false;                        // scalar
            const int r = rstart + rb + u;
            if constexpr (WFAST) filt_row(cur[u], H0[u], H1[u]);
            else hrow_edge(r, H0[u], H1[u]);
            if (r == last) {                                       // scalar
                const float b0 = 1.f - b1;
                const float B00 = col_filter<KS>(tk, KS, R, [&](int q) { return H0[(u + 1 + q) % NB]; });
                const float B01 = col_filter<KS>(tk, KS, R, [&](int q) { return H1[(u + 1 + q) % NB]; });
                float B10 = B00, B11 = B01;
                if (r1 != r0) {
                    B10 = col_filter<KS>(tk, KS, R + 1, [&](int q) { return H0[(u + 1 + q) % NB]; });
                    B11 = col_filter<KS>(tk, KS, R + 1, [&](int q) { return H1[(u + 1 + q) % NB]; });
                }
                const float t0 = NSOF_MADD(B00, a0, B01 * a1);
                const float t1 = NSOF_MADD(B10, a0, B11 * a1);
                if (live) __builtin_nontemporal_store(NSOF_MADD(t0, b0, t1 * b1), dst + (size_t)dy * wk + dx);
                dy++;
                if (dy < dy_end) {
                    row_coord(dy, r0, r1, b1);
                    last = r0 - R + KS;
                }
            }
            return true;
        };
        prep_static_slots<0, NB>(slot);
        if constexpr (WFAST) {
#pragma unroll
            for (int u = 0; u < NB; u++)
#pragma unroll
                for (int d = 0; d < ND; d++) cur[u][d] = nxt[u][d];
        }
    }
}

template <int KS>
__global__ __launch_bounds__(256) void k_prep_walk(const uint8_t* __restrict__ src, ptrdiff_t row_stride,
                                                    ptrdiff_t img_stride, int W, int H, int wk, int hk, double scale_x,
                                                    double scale_y, int seg_rows, nsof_blur_taps t, float* __restrict__ out)
{
    constexpr int R = KS / 2, NB = KS + 1, ND = (NB + 3) / 4;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int dy0 = (blockIdx.y * 4 + wave) * seg_rows;
    if (dy0 >= hk) return;                                        // wave-uniform
    const int dy_end = min(dy0 + seg_rows, hk);
    const int dxr = blockIdx.x * 64 + lane;
    const bool live = dxr < wk;
    const int dx = live ? dxr : wk - 1;
    const uint8_t* img = src + (ptrdiff_t)blockIdx.z * img_stride;
    float* dst = out + (size_t)blockIdx.z * wk * hk;
    int sx;
    float a1;
    lin_coord_x(dx, scale_x, W, sx, a1);
    const int c0 = sx, c1 = min(sx + 1, W - 1);
    const bool fast = c0 - R >= 0 && c0 - R + 4 * ND <= W && c1 == c0 + 1;
    if (__all(fast))
        prep_walk_body<KS, true>(img, row_stride, W, H, wk, hk, scale_y, dy0, dy_end, dx, live, c0, c1, true, a1, t, dst);
    else
        prep_walk_body<KS, false>(img, row_stride, W, H, wk, hk, scale_y, dy0, dy_end, dx, live, c0, c1, fast, a1, t, dst);
}

// Resampled level, two passes (kernel sizes 9 and 19: levels 2 and 3 of the reference's parameter sets).
// The destination samples only 2 source columns per destination column and 2 source rows per destination row,
// so the separable blur is evaluated only there:
//   pass A  k_prep_rows  thread <-> (source row, sampled column): KS-tap row filter -> HA [n][H][2*wk] f32
//   pass B  k_prep_cols  thread <-> destination pixel: KS-tap column filter at its 2 rows x 2 columns of HA
//                        (reflected rows), then the bilinear blend (horizontal first, as resize does).
// Both are plain thread-per-output kernels (no LDS, no barriers, full occupancy); HA is 2-4 MB per frame and
// is re-read from L2/MALL.  Same operation order as the tiled/direct kernels (bit-identical results).
constexpr int PREPA_ROWS = 8;
template <int KS>
__global__ __launch_bounds__(256) void k_prep_rows(const uint8_t* __restrict__ src, ptrdiff_t row_stride,
                                                    ptrdiff_t img_stride, int W, int H, int wk, double scale_x,
                                                    nsof_blur_taps t, float* __restrict__ HA)
{
    constexpr int R = KS / 2, ND = (KS + 3) / 4;
    const int j = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rbase = (blockIdx.y * 4 + (threadIdx.x >> 6)) * PREPA_ROWS;
    if (j >= 2 * wk || rbase >= H) return;
    int sx;
    float a;
    lin_coord_x(j >> 1, scale_x, W, sx, a);
    const int c = (j & 1) ? min(sx + 1, W - 1) : sx;
    const bool fast = c - R >= 0 && c - R + 4 * ND <= W;
    auto tk = [&](int q) { return t.k[q]; };
    const uint8_t* img = src + (ptrdiff_t)blockIdx.z * img_stride;
    float* dst = HA + ((size_t)blockIdx.z * H) * (2 * wk) + j;
#pragma unroll 2
    for (int q = 0; q < PREPA_ROWS; q++) {
        const int r = rbase + q;
        if (r >= H) break;
        const uint8_t* rowp = img + (ptrdiff_t)r * row_stride;
        float b[KS];
        if (fast) {
#pragma unroll
            for (int d = 0; d < ND; d++) {
                unsigned v;
                __builtin_memcpy(&v, rowp + (c - R) + 4 * d, 4);   // unaligned dword load
#pragma unroll
                for (int e = 0; e < 4; e++)
                    if (4 * d + e < KS) b[4 * d + e] = (float)((v >> (8 * e)) & 0xffu);
            }
        } else {
#pragma unroll
            for (int i = 0; i < KS; i++) b[i] = (float)rowp[reflect101(c - R + i, W)];
        }
        dst[(size_t)r * (2 * wk)] = row_filter<KS>(tk, KS, R, [&](int i) { return b[i]; });
    }
}

template <int KS>
__global__ __launch_bounds__(256) void k_prep_cols(const float* __restrict__ HA, int W, int H, int wk, int hk,
                                                    double scale_x, double scale_y, nsof_blur_taps t,
                                                    float* __restrict__ out)
{
    constexpr int R = KS / 2;
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dx >= wk || dy >= hk) return;
    int sx, sy;
    float a1, b1;
    lin_coord_x(dx, scale_x, W, sx, a1);
    lin_coord_y(dy, scale_y, sy, b1);
    const float a0 = 1.f - a1, b0 = 1.f - b1;
    const int r0 = clampi(sy, 0, H - 1), r1 = clampi(sy + 1, 0, H - 1);
    auto tk = [&](int q) { return t.k[q]; };
    const float2* Hz = reinterpret_cast<const float2*>(HA) + ((size_t)blockIdx.z * H) * wk + dx;
    float H0[KS + 1], H1[KS + 1];
#pragma unroll
    for (int i = 0; i <= KS; i++) {
        const float2 h = Hz[(size_t)reflect101(r0 - R + i, H) * wk];
        H0[i] = h.x;
        H1[i] = h.y;
    }
    const float B00 = col_filter<KS>(tk, KS, R, [&](int q) { return H0[q]; });
    const float B01 = col_filter<KS>(tk, KS, R, [&](int q) { return H1[q]; });
    float B10 = B00, B11 = B01;
    if (r1 != r0) {
        B10 = col_filter<KS>(tk, KS, R + 1, [&](int q) { return H0[q]; });
        B11 = col_filter<KS>(tk, KS, R + 1, [&](int q) { return H1[q]; });
    }
    const float t0 = NSOF_MADD(B00, a0, B01 * a1);
    const float t1 = NSOF_MADD(B10, a0, B11 * a1);
    out[((size_t)blockIdx.z * hk + dy) * wk + dx] = NSOF_MADD(t0, b0, t1 * b1);
}

// ---------------------------------------------------------------------------------------
// Polynomial expansion (FarnebackPolyExp).  24 B/px algorithmic: 4 read + 5 x 4 written.
//
// "Strip walker": a 256-thread block owns 256 image columns (SW output columns + NP halo on
// each side) and walks down a row segment four rows per step.
//   vertical pass   thread <-> column; the 2N+1 input rows of the column live in a register
//                   window (one coalesced dword load per thread per new row, prefetched one
//                   step ahead); r0/r1/r2 (float accumulation) go to LDS.
//   horizontal pass wave <-> row, lane <-> 4 adjacent pixels; taps come from LDS as
//                   ds_read_b128 and are reused across the 4 pixels in registers; the six
//                   moments accumulate in double exactly as the reference library does
//                   (b1,b4: double products -- exact, so written as fma; b2,b3,b5,b6: float
//                   products widened afterwards); 5 coalesced float4 stores per lane.
// ---------------------------------------------------------------------------------------
template <int N>
struct PolyGeom {
    static constexpr int NP = (N + 3) / 4 * 4;  // halo padded so LDS vectors stay 16-B aligned
    static constexpr int SW = 256 - 2 * NP;     // output columns per block
    static constexpr int NV = (2 * NP + 4) / 4; // float4 per lane per moment row
};

// FAST (opt-in, nsof_set_option(NSOF_OPT_POLYEXP_F32)): the horizontal moments accumulate in float (fma) instead of
// double -- NOT the reference library's arithmetic; results differ from the exact kernel in the last bits of R (see
// DESIGN.md for the measured end-point error).  To keep the float sums small the image is taken relative to a
// per-workgroup constant c (a constant image has zero derivatives, so the outputs do not depend on c; the
// second-derivative outputs b1*ig03 + b5*ig33 cancel their two large terms, which is where float would lose most).
template <int N, bool HET, bool FAST = false>
__global__ __launch_bounds__(256) void k_polyexp(const float* __restrict__ img, float* __restrict__ R, int W, int H,
                                                  int seg_rows, nsof_poly_taps tp,
                                                  const nsof_het_item* __restrict__ items)
{
    using G = PolyGeom<N>;
    size_t img_off, r_off;   // element offsets of this image / its expansion
    if constexpr (HET) {
        const nsof_het_item& it = items[blockIdx.z >> 1];
        const size_t which = blockIdx.z & 1;
        W = it.wk;
        H = it.hk;
        if (blockIdx.x * G::SW >= W || blockIdx.y * seg_rows >= H) return;   // block-uniform, before any barrier
        img_off = it.offI + which * (size_t)W * H;
        r_off = it.offR + which * 5 * (size_t)W * H;
    } else {
        img_off = (size_t)blockIdx.z * W * H;
        r_off = (size_t)blockIdx.z * 5 * W * H;
    }
    __shared__ __attribute__((aligned(16))) float sr[2][3][4][256];
    // The 2N double-precision taps would not fit the scalar register file next to the float taps (SGPR
    // spills cost more than the arithmetic); they live in LDS and are re-read (broadcast) once per step.
    __shared__ double stap[2][N + 1];
    __shared__ float ftap[2][N + 1];   // g, xg for the horizontal pass when N is large (see HT below)
    __shared__ float4 st[4][256];   // per-wave transpose buffer for the interleaved channel-0..3 stores

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid <= N) {
        stap[0][tid] = tp.dg[tid];
        stap[1][tid] = tp.dxxg[tid];
        ftap[0][tid] = tp.g[tid];
        ftap[1][tid] = tp.xg[tid];
    }
    const int x0 = blockIdx.x * G::SW;
    const int ys = blockIdx.y * seg_rows, ye = min(ys + seg_rows, H);
    const unsigned plane = (unsigned)W * (unsigned)H;
    // wave-uniform bases + 32-bit byte offsets: loads/stores stay in "SGPR base + VGPR offset" form
    const char* Ib = reinterpret_cast<const char*>(img + img_off);
    char* Rb = reinterpret_cast<char*>(R + r_off);
    const int xc = clampi(x0 - G::NP + tid, 0, W - 1);
    auto ld = [&](int row) {
        return *reinterpret_cast<const float*>(Ib + ((unsigned)clampi(row, 0, H - 1) * (unsigned)W + (unsigned)xc) * 4u);
    };

    // FAST: everything relative to the workgroup's first pixel
    float cref = 0.f;
    if constexpr (FAST)
        cref = *reinterpret_cast<const float*>(Ib + ((unsigned)clampi(ys, 0, H - 1) * (unsigned)W + (unsigned)clampi(x0, 0, W - 1)) * 4u);
    // register window: win[j] = I[clamp(y - N + j)][xc]
    float win[2 * N + 1];
#pragma unroll
    for (int j = 0; j <= 2 * N; j++) win[j] = ld(ys - N + j) - cref;
    float pre[4];
#pragma unroll
    for (int q = 0; q < 4; q++) pre[q] = ld(ys + 1 + N + q) - cref;

    int buf = 0;
    for (int y = ys; y < ye; y += 4, buf ^= 1) {
        float nxt[4];
#pragma unroll
        for (int q = 0; q < 4; q++) nxt[q] = ld(y + 5 + N + q) - cref;

        // ---- vertical pass: 4 rows for this thread's column
#pragma unroll
        for (int q = 0; q < 4; q++) {
            float t0 = win[N] * tp.g[0], t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int k = 1; k <= N; k++) {
                const float a = win[N - k], b = win[N + k];
                float p = a + b;
                if constexpr (FAST) {
                    t0 = fmaf(tp.g[k], p, t0);
                    t2 = fmaf(tp.xxg[k], p, t2);
                    t1 = fmaf(tp.xg[k], b - a, t1);
                } else {
                    t0 = t0 + tp.g[k] * p;
                    t2 = t2 + tp.xxg[k] * p;
                    p = b - a;
                    t1 = t1 + tp.xg[k] * p;
                }
            }
            sr[buf][0][q][tid] = t0;
            sr[buf][1][q][tid] = t1;
            sr[buf][2][q][tid] = t2;
#pragma unroll
            for (int j = 0; j < 2 * N; j++) win[j] = win[j + 1];
            win[2 * N] = pre[q];
        }
#pragma unroll
        for (int q = 0; q < 4; q++) pre[q] = nxt[q];
        __syncthreads();

        // ---- horizontal pass: wave <-> row, lane <-> 4 pixels; each moment row is consumed and its
        //      outputs stored before the next one is read (keeps the live register set small)
        const int yo = y + wave;
        const int xo = x0 + 4 * lane;
        if (4 * lane < G::SW && yo < ye && xo < W) {
            const unsigned opix = (unsigned)yo * (unsigned)W + (unsigned)xo;
            auto load_row = [&](int a, float (&v)[4 * G::NV]) {
                const float4* p4 = reinterpret_cast<const float4*>(&sr[buf][a][wave][4 * lane]);
#pragma unroll
                for (int i = 0; i < G::NV; i++) {
                    const float4 f = p4[i];
                    v[4 * i] = f.x; v[4 * i + 1] = f.y; v[4 * i + 2] = f.z; v[4 * i + 3] = f.w;
                }
            };
            double t03[4];  // b1 * ig03, shared by the xx and yy outputs
            float o0[4], o1[4], o2[4], o3[4], o4[4];
            if constexpr (FAST) {
                // float accumulation, taps in registers (xxg included), one moment row at a time
                float fg[N + 1], fxg[N + 1], fxxg[N + 1];
#pragma unroll
                for (int k = 0; k <= N; k++) {
                    fg[k] = (N > 7) ? ftap[0][k] : tp.g[k];
                    fxg[k] = (N > 7) ? ftap[1][k] : tp.xg[k];
                    fxxg[k] = tp.xxg[k];
                }
                const float i11 = (float)tp.ig11, i03 = (float)tp.ig03, i33 = (float)tp.ig33, i55 = (float)tp.ig55;
                float t03f[4];
                {
                    float v[4 * G::NV];
                    load_row(0, v);
#pragma unroll
                    for (int p = 0; p < 4; p++) {
                        const int c = G::NP + p;
                        float a1 = v[c] * fg[0], a2 = 0.f, a4 = 0.f;
#pragma unroll
                        for (int k = 1; k <= N; k++) {
                            const float hi = v[c + k], lo = v[c - k], sm = hi + lo;
                            a1 = fmaf(sm, fg[k], a1);
                            a4 = fmaf(sm, fxxg[k], a4);
                            a2 = fmaf(hi - lo, fxg[k], a2);
                        }
                        t03f[p] = a1 * i03;
                        o1[p] = a2 * i11;
                        o3[p] = fmaf(a4, i33, t03f[p]);
                    }
                }
                {
                    float v[4 * G::NV];
                    load_row(1, v);
#pragma unroll
                    for (int p = 0; p < 4; p++) {
                        const int c = G::NP + p;
                        float a3 = v[c] * fg[0], a6 = 0.f;
#pragma unroll
                        for (int k = 1; k <= N; k++) {
                            const float hi = v[c + k], lo = v[c - k];
                            a3 = fmaf(hi + lo, fg[k], a3);
                            a6 = fmaf(hi - lo, fxg[k], a6);
                        }
                        o0[p] = a3 * i11;
                        o4[p] = a6 * i55;
                    }
                }
                {
                    float v[4 * G::NV];
                    load_row(2, v);
#pragma unroll
                    for (int p = 0; p < 4; p++) {
                        const int c = G::NP + p;
                        float a5 = v[c] * fg[0];
#pragma unroll
                        for (int k = 1; k <= N; k++) a5 = fmaf(v[c + k] + v[c - k], fg[k], a5);
                        o2[p] = fmaf(a5, i33, t03f[p]);
                    }
                }
            } else {
            // Large radii: 3(N+1) float + 2N double taps exceed the scalar register file (the spills cost more than
            // the arithmetic), so the horizontal pass takes its float taps from LDS into VGPRs as well.
            constexpr bool HT = N > 7;
            float hg[N + 1], hxg[N + 1];
#pragma unroll
            for (int k = 0; k <= N; k++) {
                hg[k] = HT ? ftap[0][k] : tp.g[k];
                hxg[k] = HT ? ftap[1][k] : tp.xg[k];
            }
            {
                float v[4 * G::NV];
                double dg[N + 1], dxxg[N + 1];
#pragma unroll
                for (int k = 1; k <= N; k++) {
                    dg[k] = stap[0][k];
                    dxxg[k] = stap[1][k];
                }
                load_row(0, v);
#pragma unroll
                for (int p = 0; p < 4; p++) {
                    const int c = G::NP + p;
                    double a1 = (double)(v[c] * hg[0]), a2 = 0, a4 = 0;
#pragma unroll
                    for (int k = 1; k <= N; k++) {
                        const float hi = v[c + k], lo = v[c - k];
                        const double tg = (double)(hi + lo);
                        a1 = fma(tg, dg[k], a1);     // product of two float-valued doubles is exact
                        a4 = fma(tg, dxxg[k], a4);
                        a2 += (double)((hi - lo) * hxg[k]);
                    }
                    t03[p] = a1 * tp.ig03;
                    o1[p] = (float)(a2 * tp.ig11);
                    o3[p] = (float)(t03[p] + a4 * tp.ig33);
                }
            }
            {
                float v[4 * G::NV];
                load_row(1, v);
#pragma unroll
                for (int p = 0; p < 4; p++) {
                    const int c = G::NP + p;
                    double a3 = (double)(v[c] * hg[0]), a6 = 0;
#pragma unroll
                    for (int k = 1; k <= N; k++) {
                        const float hi = v[c + k], lo = v[c - k];
                        a3 += (double)((hi + lo) * hg[k]);
                        a6 += (double)((hi - lo) * hxg[k]);
                    }
                    o0[p] = (float)(a3 * tp.ig11);
                    o4[p] = (float)(a6 * tp.ig55);
                }
            }
            {
                float v[4 * G::NV];
                load_row(2, v);
#pragma unroll
                for (int p = 0; p < 4; p++) {
                    const int c = G::NP + p;
                    double a5 = (double)(v[c] * hg[0]);
#pragma unroll
                    for (int k = 1; k <= N; k++) a5 += (double)((v[c + k] + v[c - k]) * hg[k]);
                    o2[p] = (float)(t03[p] + a5 * tp.ig33);
                }
            }
            }   // !FAST
            // channel 4 of the lane's 4 pixels: one 16-B store
            float* c4 = reinterpret_cast<float*>(Rb) + 4u * plane + opix;
            if ((W & 3) == 0) {
                nsof_store_stream4(c4, o4[0], o4[1], o4[2], o4[3]);
            } else {
#pragma unroll
                for (int p = 0; p < 4; p++)
                    if (xo + p < W) c4[p] = o4[p];
            }
            // channels 0-3: a lane holds 4 consecutive pixels x 16 B; stored as is, one instruction would write 16 B
            // per lane at a 64-B stride.  Transpose through LDS (per wave) so that every store instruction writes
            // 64 consecutive pixels = 1 KiB contiguous.
#pragma unroll
            // (swizzled within each lane's 4 slots: unswizzled, the 8 lanes a ds_write_b128 serves per LDS cycle hit
            //  only 2 of the 8 bank groups -- PMC: 63 % of this kernel's LDS cycles were bank conflicts)
            for (int p = 0; p < 4; p++) st[wave][4 * lane + (p ^ ((lane >> 1) & 3))] = make_float4(o0[p], o1[p], o2[p], o3[p]);
        }
        {
            // every lane of the wave takes part (lanes beyond the strip read slots nobody wrote, and do not store)
            const int yo2 = y + wave;
            float4* q4 = reinterpret_cast<float4*>(Rb) + (unsigned)yo2 * (unsigned)W + (unsigned)x0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int px = 64 * k + lane;   // pixel within the strip row
                const float4 v = st[wave][(px & ~3) | ((px & 3) ^ ((px >> 3) & 3))];   // pixel px sits in lane px/4's slot
                if (px < G::SW && yo2 < ye && x0 + px < W) nsof_store_stream4(reinterpret_cast<float*>(q4 + px), v.x, v.y, v.z, v.w);
            }
        }
        // no second barrier: the next step writes the other LDS buffer (st is private to a wave)
    }
}

// ---------------------------------------------------------------------------------------
// Role-specialised strip walker (the default for the exact arithmetic): the same two passes, same arithmetic and order
// as k_polyexp, run by different waves.  A workgroup has 8 waves: waves 0-3 (thread <-> column) run the vertical pass
// of step t+1 while waves 4-7 (wave <-> row, lane <-> 4 pixels) run the horizontal pass of step t on the other half of
// the double-buffered moment rows; one barrier per step.  Neither role carries the other's registers across its
// pass (the column window of 2N+1 rows on one side, the moment window and the double taps on the other), so the
// taps stay in scalar registers and radius 10 fits 4 waves per SIMD where the single-role kernel spilled at 256.
// ---------------------------------------------------------------------------------------
// U8 (the full-resolution level): the level image is not read from memory but formed in the vertical pass from the
// 8-bit frame itself -- the 3 x 3 [k1 k0 k1] smoothing of k_prep_same3_vec, operation for operation -- so the
// pyramid kernel of level 0 and the 8 B/px its image costs (written there, read here) disappear; the vertical-pass
// waves have the issue slots for it (187 of their step's ~500 instruction slots were used).
struct PolyU8 {
    const uint8_t* src0;   // images [0, nsplit) at src0 + z * img_stride, the others at src1 + (z - nsplit) * img_stride
    const uint8_t* src1;
    long long row_stride, img_stride;
    int nsplit;
    float k0, k1;          // centre and side tap
};
template <int N, bool HET, bool U8 = false>
__global__ __launch_bounds__(512) void k_polyexp_rs(const float* __restrict__ img, float* __restrict__ R, int W, int H,
                                                     int seg_rows, nsof_poly_taps tp,
                                                     const nsof_het_item* __restrict__ items, PolyU8 u8 = PolyU8{})
{
    using G = PolyGeom<N>;
    size_t img_off, r_off;   // element offsets of this image / its expansion
    const uint8_t* sb = nullptr;   // U8: this image's frame
    long long srs = 0;
    if constexpr (HET) {
        const nsof_het_item& it = items[blockIdx.z >> 1];
        const size_t which = blockIdx.z & 1;
        W = it.wk;
        H = it.hk;
        if (blockIdx.x * G::SW >= W || blockIdx.y * seg_rows >= H) return;   // block-uniform, before any barrier
        img_off = it.offI + which * (size_t)W * H;
        r_off = it.offR + which * 5 * (size_t)W * H;
        if constexpr (U8) {
            sb = it.src[which];
            srs = it.src_stride[which];
        }
    } else {
        img_off = (size_t)blockIdx.z * W * H;
        r_off = (size_t)blockIdx.z * 5 * W * H;
        if constexpr (U8) {
            const int z = blockIdx.z;
            sb = z < u8.nsplit ? u8.src0 + (ptrdiff_t)z * u8.img_stride : u8.src1 + (ptrdiff_t)(z - u8.nsplit) * u8.img_stride;
            srs = u8.row_stride;
        }
    }
    __shared__ __attribute__((aligned(16))) float sr[2][3][4][256];
    __shared__ float4 st[4][256];   // per-wave transpose buffer for the interleaved channel-0..3 stores

    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;   // within the role
    const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
    const int x0 = blockIdx.x * G::SW;
    const int ys = blockIdx.y * seg_rows, ye = min(ys + seg_rows, H);
    const int nsteps = (ye - ys + 3) / 4;
    const unsigned plane = (unsigned)W * (unsigned)H;

    if (role == 0) {
        // ---- vertical pass: 4 rows per step for this thread's column
        const char* Ib = reinterpret_cast<const char*>(img + img_off);
        const int xc = clampi(x0 - G::NP + tid, 0, W - 1);
        auto ld = [&](int row) {
            return *reinterpret_cast<const float*>(Ib + ((unsigned)clampi(row, 0, H - 1) * (unsigned)W + (unsigned)xc) * 4u);
        };
        // U8: I[rc][xc] from the frame.  The rows are asked for in order, so the row-filtered values of rows rc - 1, rc,
        // rc + 1 (reflected at the image border like the pyramid kernel's) are kept and one new row is filtered per
        // new rc; fetch3() only issues the loads of a row's three bytes (four rows ahead, like the float loads).
        const int xl = U8 ? reflect101(xc - 1, W) : 0, xr = U8 ? reflect101(xc + 1, W) : 0;
        struct Raw3 {
            unsigned l, c, r;
        };
        auto fetch3 = [&](int srow) {   // srow: a row of the frame
            const uint8_t* rp = sb + (ptrdiff_t)srow * srs;
            return Raw3{rp[xl], rp[xc], rp[xr]};
        };
        auto hval = [&](const Raw3& q) { return NSOF_MADD((float)q.l + (float)q.r, u8.k1, (float)q.c * u8.k0); };
        int rc_cur = 0;
        float hm = 0.f, h0 = 0.f, hp = 0.f, icur = 0.f;
        auto below = [&](int row) { return reflect101(clampi(row, 0, H - 1) + 1, H); };   // the frame row under clamp(row)
        auto advance = [&](int row, const Raw3& under) {   // I[clamp(row)][xc]; under = fetch3(below(row))
            const int rc = clampi(row, 0, H - 1);
            if (rc != rc_cur) {   // block-uniform; rows come in order: rc == rc_cur + 1
                hm = h0;
                h0 = hp;
                hp = hval(under);
                icur = NSOF_MADD(hm + hp, u8.k1, h0 * u8.k0);
                rc_cur = rc;
            }
            return icur;
        };
        float win[2 * N + 1];   // win[j] = I[clamp(y - N + j)][xc]
        float pre[4];
        Raw3 praw[4];
        if constexpr (U8) {
            rc_cur = clampi(ys - N, 0, H - 1);
            hm = hval(fetch3(reflect101(rc_cur - 1, H)));
            h0 = hval(fetch3(rc_cur));
            hp = hval(fetch3(reflect101(rc_cur + 1, H)));
            icur = NSOF_MADD(hm + hp, u8.k1, h0 * u8.k0);
            win[0] = icur;
#pragma unroll
            for (int j = 1; j <= 2 * N; j++) win[j] = advance(ys - N + j, fetch3(below(ys - N + j)));
#pragma unroll
            for (int q = 0; q < 4; q++) pre[q] = advance(ys + 1 + N + q, fetch3(below(ys + 1 + N + q)));
        } else {
#pragma unroll
            for (int j = 0; j <= 2 * N; j++) win[j] = ld(ys - N + j);
#pragma unroll
            for (int q = 0; q < 4; q++) pre[q] = ld(ys + 1 + N + q);
        }
        for (int t = 0; t <= nsteps; t++) {
            if (t < nsteps) {
                const int y = ys + 4 * t, buf = t & 1;
                float nxt[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    if constexpr (U8) praw[q] = fetch3(below(y + 5 + N + q));
                    else nxt[q] = ld(y + 5 + N + q);
                }
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    float t0 = win[N] * tp.g[0], t1 = 0.f, t2 = 0.f;
#pragma unroll
                    for (int k = 1; k <= N; k++) {
                        const float a = win[N - k], b = win[N + k];
                        float p = a + b;
                        t0 = t0 + tp.g[k] * p;
                        t2 = t2 + tp.xxg[k] * p;
                        p = b - a;
                        t1 = t1 + tp.xg[k] * p;
                    }
                    sr[buf][0][q][tid] = t0;
                    sr[buf][1][q][tid] = t1;
                    sr[buf][2][q][tid] = t2;
#pragma unroll
                    for (int j = 0; j < 2 * N; j++) win[j] = win[j + 1];
                    win[2 * N] = pre[q];
                }
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    if constexpr (U8) pre[q] = advance(y + 5 + N + q, praw[q]);
                    else pre[q] = nxt[q];
                }
            }
            __syncthreads();
        }
        return;
    }

    // ---- horizontal pass: wave <-> row, lane <-> 4 pixels, one step behind the vertical pass
    char* Rb = reinterpret_cast<char*>(R + r_off);
    for (int t = 0; t <= nsteps; t++) {
        if (t >= 1) {
            const int y = ys + 4 * (t - 1), buf = (t - 1) & 1;
            const int yo = y + wave;
            const int xo = x0 + 4 * lane;
            if (4 * lane < G::SW && yo < ye && xo < W) {
                const unsigned opix = (unsigned)yo * (unsigned)W + (unsigned)xo;
                auto load_row = [&](int a, float (&v)[4 * G::NV]) {
                    const float4* p4 = reinterpret_cast<const float4*>(&sr[buf][a][wave][4 * lane]);
#pragma unroll
                    for (int i = 0; i < G::NV; i++) {
                        const float4 f = p4[i];
                        v[4 * i] = f.x; v[4 * i + 1] = f.y; v[4 * i + 2] = f.z; v[4 * i + 3] = f.w;
                    }
                };
                double t03[4];  // b1 * ig03, shared by the xx and yy outputs
                float o0[4], o1[4], o2[4], o3[4], o4[4];
                {
                    float v[4 * G::NV];
                    load_row(0, v);
#pragma unroll
                    for (int p = 0; p < 4; p++) {
                        const int c = G::NP + p;
                        double a1 = (double)(v[c] * tp.g[0]), a2 = 0, a4 = 0;
#pragma unroll
                        for (int k = 1; k <= N; k++) {
                            const float hi = v[c + k], lo = v[c - k];
                            const double tg = (double)(hi + lo);
                            a1 = fma(tg, tp.dg[k], a1);     // product of two float-valued doubles is exact
                            a4 = fma(tg, tp.dxxg[k], a4);
                            a2 += (double)((hi - lo) * tp.xg[k]);
                        }
                        t03[p] = a1 * tp.ig03;
                        o1[p] = (float)(a2 * tp.ig11);
                        o3[p] = (float)(t03[p] + a4 * tp.ig33);
                    }
                }
                {
                    float v[4 * G::NV];
                    load_row(1, v);
#pragma unroll
                    for (int p = 0; p < 4; p++) {
                        const int c = G::NP + p;
                        double a3 = (double)(v[c] * tp.g[0]), a6 = 0;
#pragma unroll
                        for (int k = 1; k <= N; k++) {
                            const float hi = v[c + k], lo = v[c - k];
                            a3 += (double)((hi + lo) * tp.g[k]);
                            a6 += (double)((hi - lo) * tp.xg[k]);
                        }
                        o0[p] = (float)(a3 * tp.ig11);
                        o4[p] = (float)(a6 * tp.ig55);
                    }
                }
                {
                    float v[4 * G::NV];
                    load_row(2, v);
#pragma unroll
                    for (int p = 0; p < 4; p++) {
                        const int c = G::NP + p;
                        double a5 = (double)(v[c] * tp.g[0]);
#pragma unroll
                        for (int k = 1; k <= N; k++) a5 += (double)((v[c + k] + v[c - k]) * tp.g[k]);
                        o2[p] = (float)(t03[p] + a5 * tp.ig33);
                    }
                }
                float* c4 = reinterpret_cast<float*>(Rb) + 4u * plane + opix;
                if ((W & 3) == 0) {
                    nsof_store_stream4(c4, o4[0], o4[1], o4[2], o4[3]);
                } else {
#pragma unroll
                    for (int p = 0; p < 4; p++)
                        if (xo + p < W) c4[p] = o4[p];
                }
#pragma unroll
                for (int p = 0; p < 4; p++) st[wave][4 * lane + (p ^ ((lane >> 1) & 3))] = make_float4(o0[p], o1[p], o2[p], o3[p]);
            }
            {
                float4* q4 = reinterpret_cast<float4*>(Rb) + (unsigned)yo * (unsigned)W + (unsigned)x0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int px = 64 * k + lane;   // pixel within the strip row
                    const float4 v = st[wave][(px & ~3) | ((px & 3) ^ ((px >> 3) & 3))];
                    if (px < G::SW && yo < ye && x0 + px < W) nsof_store_stream4(reinterpret_cast<float*>(q4 + px), v.x, v.y, v.z, v.w);
                }
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------
// FarnebackUpdateMatrices: one thread per pixel.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_update_matrices(const float* __restrict__ R0b, const float* __restrict__ R1b,
                                                          size_t pair_stride, const float* __restrict__ flow,
                                                          int W, int H, float* __restrict__ M)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const size_t plane = (size_t)W * H;
    const float* R0 = R0b + (size_t)blockIdx.z * pair_stride;
    const float* R1 = R1b + (size_t)blockIdx.z * pair_stride;
    const size_t pix = (size_t)y * W + x;
    const float2 d = reinterpret_cast<const float2*>(flow)[(size_t)blockIdx.z * plane + pix];
    const float dx = d.x, dy = d.y;
    float fx = x + dx, fy = y + dy;
    const int x1 = floor_f(fx), y1 = floor_f(fy);
    fx -= x1;
    fy -= y1;
    float r2, r3, r4, r5, r6;
    if ((unsigned)x1 < (unsigned)(W - 1) && (unsigned)y1 < (unsigned)(H - 1)) {
        const float a00 = (1.f - fx) * (1.f - fy), a01 = fx * (1.f - fy), a10 = (1.f - fx) * fy, a11 = fx * fy;
        const float4* q = reinterpret_cast<const float4*>(R1) + (size_t)y1 * W + x1;
        const float* c4 = R1 + 4 * plane + (size_t)y1 * W + x1;
        const float4 t0 = q[0], t1 = q[1], b0 = q[W], b1 = q[W + 1];
        r2 = a00 * t0.x + a01 * t1.x + a10 * b0.x + a11 * b1.x;
        r3 = a00 * t0.y + a01 * t1.y + a10 * b0.y + a11 * b1.y;
        r4 = a00 * t0.z + a01 * t1.z + a10 * b0.z + a11 * b1.z;
        r5 = a00 * t0.w + a01 * t1.w + a10 * b0.w + a11 * b1.w;
        r6 = a00 * c4[0] + a01 * c4[1] + a10 * c4[W] + a11 * c4[W + 1];
        const float4 z = reinterpret_cast<const float4*>(R0)[pix];
        r4 = (z.z + r4) * 0.5f;
        r5 = (z.w + r5) * 0.5f;
        r6 = (R0[4 * plane + pix] + r6) * 0.25f;
        r2 = (z.x - r2) * 0.5f;
        r3 = (z.y - r3) * 0.5f;
    } else {
        const float4 z = reinterpret_cast<const float4*>(R0)[pix];
        r4 = z.z;
        r5 = z.w;
        r6 = R0[4 * plane + pix] * 0.5f;
        r2 = (z.x - 0.f) * 0.5f;
        r3 = (z.y - 0.f) * 0.5f;
    }
    r2 += r4 * dy + r6 * dx;
    r3 += r6 * dy + r5 * dx;
    if ((unsigned)(x - 5) >= (unsigned)(W - 10) || (unsigned)(y - 5) >= (unsigned)(H - 10)) {
        const float border[5] = {0.14f, 0.14f, 0.4472f, 0.4472f, 0.4472f};
        auto bw = [&](int i) { return i == 0 || i == 1 ? border[0] : border[2]; };
        const float scale = (x < 5 ? bw(x) : 1.f) * (x >= W - 5 ? bw(W - x - 1) : 1.f) * (y < 5 ? bw(y) : 1.f) *
                            (y >= H - 5 ? bw(H - y - 1) : 1.f);
        r2 *= scale; r3 *= scale; r4 *= scale; r5 *= scale; r6 *= scale;
    }
    float* Mz = M + (size_t)blockIdx.z * 5 * plane + pix;
    Mz[0] = r4 * r4 + r6 * r6;
    Mz[plane] = (r4 + r5) * r6;
    Mz[2 * plane] = r5 * r5 + r6 * r6;
    Mz[3 * plane] = r4 * r2 + r6 * r3;
    Mz[4 * plane] = r6 * r2 + r5 * r3;
}

// ---------------------------------------------------------------------------------------
// FarnebackUpdateFlow_Blur: (2m+1)^2 box sums of the 5 planes of M + per-pixel 2x2 solve.
//
// Strip walker over the full image height (the column sums are a running sum from row 0:
// each row adds double(float(M[y+m] - M[y-m-1])) -- the float rounding of the difference is
// part of the reference arithmetic and is reproduced).  thread <-> column keeps the 5
// column sums in registers as doubles; 4 rows per step go to LDS; wave <-> row, lane <-> 4
// pixels forms the row sums (first pixel direct, then sliding) and solves.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_blur_solve(const float* __restrict__ M, int W, int H, int m, int block_size,
                                                     float* __restrict__ flow)
{
    __shared__ double sv[4][5][256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int SW = (256 - 2 * m) & ~3;  // multiple of 4: every lane owns 4 whole pixels
    const int x0 = blockIdx.x * SW;
    const size_t plane = (size_t)W * H;
    const float* Mz = M + (size_t)blockIdx.z * 5 * plane;
    float2* fz = reinterpret_cast<float2*>(flow) + (size_t)blockIdx.z * plane;
    const int xc = clampi(x0 - m + tid, 0, W - 1);
    const float* Mc = Mz + xc;
    const double scale = 1. / (block_size * block_size);

    double vs[5];
#pragma unroll
    for (int c = 0; c < 5; c++) {
        vs[c] = (double)(Mc[c * plane] * (float)(m + 2));  // float product, as "srow0[x]*(m+2)"
        for (int y = 1; y < m; y++) vs[c] += (double)Mc[c * plane + (size_t)min(y, H - 1) * W];
    }
    float pa[4][5], pb[4][5];
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int c = 0; c < 5; c++) {
            pa[q][c] = Mc[c * plane + (size_t)min(q + m, H - 1) * W];
            pb[q][c] = Mc[c * plane + (size_t)max(q - m - 1, 0) * W];
        }

    for (int y = 0; y < H; y += 4) {
        float na[4][5], nb[4][5];
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int c = 0; c < 5; c++) {
                na[q][c] = Mc[c * plane + (size_t)min(y + 4 + q + m, H - 1) * W];
                nb[q][c] = Mc[c * plane + (size_t)max(y + 4 + q - m - 1, 0) * W];
            }
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int c = 0; c < 5; c++) {
                const float d = pa[q][c] - pb[q][c];
                vs[c] += (double)d;
                sv[q][c][tid] = vs[c];
            }
        __syncthreads();
        const int yo = y + wave, xo = x0 + 4 * lane;
        if (4 * lane < SW && yo < H && xo < W) {
            double g[5];
            float2 o[4];
#pragma unroll
            for (int p = 0; p < 4; p++) {
                if (p == 0) {
#pragma unroll
                    for (int c = 0; c < 5; c++) {
                        double s = 0;
                        for (int j = 0; j <= 2 * m; j++) s += sv[wave][c][4 * lane + j];
                        g[c] = s;
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < 5; c++)
                        g[c] += sv[wave][c][4 * lane + p + 2 * m] - sv[wave][c][4 * lane + p - 1];
                }
                const double g11 = g[0] * scale, g12 = g[1] * scale, g22 = g[2] * scale;
                const double h1 = g[3] * scale, h2 = g[4] * scale;
                const double idet = nsof_recip_normal(g11 * g22 - g12 * g12 + 1e-3);
                o[p].x = (float)((g11 * h2 - g12 * h1) * idet);
                o[p].y = (float)((g22 * h1 - g12 * h2) * idet);
            }
            float2* dst = fz + (size_t)yo * W + xo;
#pragma unroll
            for (int p = 0; p < 4; p++)
                if (4 * lane + p < SW && xo + p < W) dst[p] = o[p];
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int c = 0; c < 5; c++) {
                pa[q][c] = na[q][c];
                pb[q][c] = nb[q][c];
            }
    }
}

// ---------------------------------------------------------------------------------------
// FarnebackUpdateFlow_Blur in the reference library's EXACT summation order (NSOF_OPT_EXACT_ROWSUMS).
//
// The library forms the (2m+1)-wide ROW sums as ONE running sum along the whole image row, in double:
// g += vsum[x+m] - vsum[x-m-1] for x = 0..W-1.  The production kernels sum each pixel's window directly -- the same
// numbers to about 1e-16 relative.  Where the 2x2 system is rank deficient (straight edges, flat areas: g11*g22 -
// g12^2 cancels down to the 1e-3 regulariser) those last bits decide the flow's 4th decimal, so on real footage a few
// pixels per frame differ from the library by 1e-4..1e-3.  This pair of kernels reproduces the
// library's order bit for bit at roughly half the speed: the column sums go to HBM (transposed, 40 B/px) and a
// thread walks each image row from left to right.
// ---------------------------------------------------------------------------------------
// thread <-> column: vertical running sums of the 5 planes of M -> VT [n][5][W][H] (transposed: row index fastest)
__global__ __launch_bounds__(256) void k_blur_colsum(const float* __restrict__ M, int W, int H, int m,
                                                      double* __restrict__ VT)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= W) return;
    const size_t plane = (size_t)W * H;
    const float* Mc = M + (size_t)blockIdx.z * 5 * plane + x;
    double* V = VT + (size_t)blockIdx.z * 5 * plane + (size_t)x * H;
    double vs[5];
#pragma unroll
    for (int c = 0; c < 5; c++) {
        vs[c] = (double)(Mc[c * plane] * (float)(m + 2));   // float product, as "srow0[x]*(m+2)"
        for (int y = 1; y < m; y++) vs[c] += (double)Mc[c * plane + (size_t)min(y, H - 1) * W];
    }
    for (int y = 0; y < H; y++) {
        const size_t ra = (size_t)min(y + m, H - 1) * W, rb = (size_t)max(y - m - 1, 0) * W;
#pragma unroll
        for (int c = 0; c < 5; c++) {
            const float d = Mc[c * plane + ra] - Mc[c * plane + rb];   // rounded to float before it is added
            vs[c] += (double)d;
            V[c * plane + y] = vs[c];
        }
    }
}

// thread <-> row: the library's running row sums + the 2x2 solve, left to right
__global__ __launch_bounds__(64) void k_blur_rowsolve(const double* __restrict__ VT, int W, int H, int m, int block_size,
                                                       float* __restrict__ flow)
{
    const int y = blockIdx.x * 64 + threadIdx.x;
    if (y >= H) return;
    const size_t plane = (size_t)W * H;
    const double* V = VT + (size_t)blockIdx.z * 5 * plane + y;          // V[c*plane + x*H]
    float2* fz = reinterpret_cast<float2*>(flow) + (size_t)blockIdx.z * plane + (size_t)y * W;
    const double scale = 1. / (block_size * block_size);
    auto at = [&](int c, int x) { return V[c * plane + (size_t)clampi(x, 0, W - 1) * H]; };
    double g[5];
#pragma unroll
    for (int c = 0; c < 5; c++) {
        g[c] = at(c, 0) * (m + 2);
        for (int x = 1; x < m; x++) g[c] += at(c, x);
    }
    for (int x = 0; x < W; x++) {
#pragma unroll
        for (int c = 0; c < 5; c++) g[c] += at(c, x + m) - at(c, x - m - 1);
        const double g11 = g[0] * scale, g12 = g[1] * scale, g22 = g[2] * scale;
        const double h1 = g[3] * scale, h2 = g[4] * scale;
        const double idet = nsof_recip_normal(g11 * g22 - g12 * g12 + 1e-3);
        fz[x] = make_float2((float)((g11 * h2 - g12 * h1) * idet), (float)((g22 * h1 - g12 * h2) * idet));
    }
}

// ---------------------------------------------------------------------------------------
// Coarse-to-fine flow resample: resize(prevFlow, INTER_LINEAR) then "flow *= 1/pyr_scale".
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_flow_upsample(const float* __restrict__ src, int sw, int sh,
                                                        float* __restrict__ dst, int dw, int dh, double scale_x,
                                                        double scale_y, float mul)
{
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dx >= dw || dy >= dh) return;
    int sx, sy;
    float a1, b1;
    lin_coord_x(dx, scale_x, sw, sx, a1);
    lin_coord_y(dy, scale_y, sy, b1);
    const float a0 = 1.f - a1, b0 = 1.f - b1;
    const int c1 = min(sx + 1, sw - 1);
    const int r0 = clampi(sy, 0, sh - 1), r1 = clampi(sy + 1, 0, sh - 1);
    const float2* S = reinterpret_cast<const float2*>(src) + (size_t)blockIdx.z * sw * sh;
    const float2 p00 = S[(size_t)r0 * sw + sx], p01 = S[(size_t)r0 * sw + c1];
    const float2 p10 = S[(size_t)r1 * sw + sx], p11 = S[(size_t)r1 * sw + c1];
    float2 o;
    {
        const float t0 = NSOF_MADD(p00.x, a0, p01.x * a1), t1 = NSOF_MADD(p10.x, a0, p11.x * a1);
        o.x = NSOF_MADD(t0, b0, t1 * b1) * mul;
    }
    {
        const float t0 = NSOF_MADD(p00.y, a0, p01.y * a1), t1 = NSOF_MADD(p10.y, a0, p11.y * a1);
        o.y = NSOF_MADD(t0, b0, t1 * b1) * mul;
    }
    reinterpret_cast<float2*>(dst)[((size_t)blockIdx.z * dh + dy) * dw + dx] = o;
}

// Upsampling variant: a thread produces a 2x2 block of destination pixels.  When dst >= src in both directions,
// consecutive destination coordinates map to source coordinates at most one apart, so the block's 4 pixels
// sample from a 3x3 source neighbourhood: 9 float2 loads + 2 float4 stores per 4 pixels instead of 16 + 4 (the L1
// serves 4 lanes per cycle per instruction, whatever its width).  Same arithmetic per pixel as k_flow_upsample.
template <bool HET>
__global__ __launch_bounds__(256) void k_flow_upsample2x2(const float* __restrict__ src, int sw, int sh,
                                                           float* __restrict__ dst, int dw, int dh, double scale_x,
                                                           double scale_y, float mul,
                                                           const nsof_het_item* __restrict__ items)
{
    const int bx = blockIdx.x * 64 + (threadIdx.x & 63), by = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int dx0 = 2 * bx, dy0 = 2 * by;
    size_t s_off, d_off;   // float2 offsets of this field in src / dst
    if constexpr (HET) {
        const nsof_het_item& it = items[blockIdx.z];
        sw = it.pw; sh = it.ph; dw = it.wk; dh = it.hk;
        s_off = it.offFc;
        d_off = it.offF;
        if (dx0 >= dw || dy0 >= dh) return;
        if (sw == 0) {   // the item's coarsest level: its incoming flow is zero
            float2* D = reinterpret_cast<float2*>(dst) + d_off;
            for (int i = 0; i < 2 && dy0 + i < dh; i++)
                for (int j = 0; j < 2 && dx0 + j < dw; j++) D[(size_t)(dy0 + i) * dw + dx0 + j] = make_float2(0.f, 0.f);
            return;
        }
        scale_x = 1. / ((double)dw / sw);
        scale_y = 1. / ((double)dh / sh);
    } else {
        s_off = (size_t)blockIdx.z * sw * sh;
        d_off = (size_t)blockIdx.z * dw * dh;
        if (dx0 >= dw || dy0 >= dh) return;
    }
    int sx[2], sy[2];
    float a1[2], b1[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        lin_coord_x(min(dx0 + i, dw - 1), scale_x, sw, sx[i], a1[i]);
        lin_coord_y(min(dy0 + i, dh - 1), scale_y, sy[i], b1[i]);
    }
    const float2* S = reinterpret_cast<const float2*>(src) + s_off;
    // source columns sx[0]+{0,1,2} and rows sy[0]+{0,1,2}, clamped like the per-pixel kernel clamps them
    float2 v[3][3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const float2* row = S + (size_t)clampi(sy[0] + r, 0, sh - 1) * sw;
#pragma unroll
        for (int c = 0; c < 3; c++) v[r][c] = row[min(sx[0] + c, sw - 1)];
    }
    float2* D = reinterpret_cast<float2*>(dst) + d_off;
#pragma unroll
    for (int i = 0; i < 2; i++) {          // destination row dy0 + i
        if (dy0 + i >= dh) break;
        const int ro = sy[i] - sy[0];       // 0 or 1
        const float bb1 = b1[i], bb0 = 1.f - bb1;
        float2 o[2];
#pragma unroll
        for (int j = 0; j < 2; j++) {      // destination column dx0 + j
            const int co = sx[j] - sx[0];   // 0 or 1
            const float aa1 = a1[j], aa0 = 1.f - aa1;
            const float2 p00 = ro ? (co ? v[1][1] : v[1][0]) : (co ? v[0][1] : v[0][0]);
            const float2 p01 = ro ? (co ? v[1][2] : v[1][1]) : (co ? v[0][2] : v[0][1]);
            const float2 p10 = ro ? (co ? v[2][1] : v[2][0]) : (co ? v[1][1] : v[1][0]);
            const float2 p11 = ro ? (co ? v[2][2] : v[2][1]) : (co ? v[1][2] : v[1][1]);
            {
                const float t0 = NSOF_MADD(p00.x, aa0, p01.x * aa1), t1 = NSOF_MADD(p10.x, aa0, p11.x * aa1);
                o[j].x = NSOF_MADD(t0, bb0, t1 * bb1) * mul;
            }
            {
                const float t0 = NSOF_MADD(p00.y, aa0, p01.y * aa1), t1 = NSOF_MADD(p10.y, aa0, p11.y * aa1);
                o[j].y = NSOF_MADD(t0, bb0, t1 * bb1) * mul;
            }
        }
        float2* drow = D + (size_t)(dy0 + i) * dw + dx0;
        if (dx0 + 1 < dw && (dw & 1) == 0 && (!HET || (d_off & 1) == 0))
            nsof_store_stream4(reinterpret_cast<float*>(drow), o[0].x, o[0].y, o[1].x, o[1].y);
        else {
            drow[0] = o[0];
            if (dx0 + 1 < dw) drow[1] = o[1];
        }
    }
}

// Upsampling as a row walk: a lane owns 2 adjacent destination columns (one 16-B store per row, so that a store
// instruction of the wave writes 1 KiB contiguous) and walks down a segment of destination rows.  The (at most 3)
// source columns its pixels sample are loaded once per SOURCE row, blended horizontally once, and reused by the
// 2-3 destination rows that sample that source row.  Per-pixel arithmetic and its order are those of
// k_flow_upsample.
constexpr int UPW_SEG = 32;
__global__ __launch_bounds__(256) void k_flow_upsample_walk(const float* __restrict__ src, int sw, int sh,
                                                             float* __restrict__ dst, int dw, int dh, double scale_x,
                                                             double scale_y, float mul)
{
    constexpr int NPL = 2, NV = NPL + 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // the 4 waves of a block take adjacent column chunks of the SAME rows: 4 KiB contiguous per row and block
    const int dx0 = ((blockIdx.x * 4 + wave) * 64 + lane) * NPL;
    const int y0 = blockIdx.y * UPW_SEG;
    if (y0 >= dh || (blockIdx.x * 4 + wave) * 64 * NPL >= dw) return;   // wave-uniform
    const int y1 = min(y0 + UPW_SEG, dh);
    const bool live = dx0 < dw;
    int base = 0, i0[NPL], i1[NPL];
    float a0[NPL], a1[NPL];
#pragma unroll
    for (int i = 0; i < NPL; i++) {
        int sx;
        lin_coord_x(min(dx0 + i, dw - 1), scale_x, sw, sx, a1[i]);
        a0[i] = 1.f - a1[i];
        if (i == 0) base = sx;
        i0[i] = sx - base;                                    // 0..NPL-1 when upsampling
        i1[i] = min(sx + 1, sw - 1) - base;                   // 0..NPL
    }
    const float2* S = reinterpret_cast<const float2*>(src) + (size_t)blockIdx.z * sw * sh;
    float2* D = reinterpret_cast<float2*>(dst) + (size_t)blockIdx.z * dw * dh;
    auto pick = [](const float2 (&V)[NV], int k) {
        float2 r = V[0];
#pragma unroll
        for (int q = 1; q < NV; q++) r = k == q ? V[q] : r;
        return r;
    };
    // horizontal blend of source row r at the lane's destination columns
    auto hrow = [&](int r, float2 (&h)[NPL]) {
        const float2* row = S + (size_t)r * sw;
        float2 V[NV];
#pragma unroll
        for (int k = 0; k < NV; k++) V[k] = row[min(base + k, sw - 1)];
#pragma unroll
        for (int i = 0; i < NPL; i++) {
            const float2 p0 = pick(V, i0[i]), p1 = pick(V, i1[i]);
            h[i].x = NSOF_MADD(p0.x, a0[i], p1.x * a1[i]);
            h[i].y = NSOF_MADD(p0.y, a0[i], p1.y * a1[i]);
        }
    };
    float2 hA[NPL], hB[NPL];
    int ra = -1, rb = -1;
    for (int dy = y0; dy < y1; dy++) {
        int sy;
        float b1;
        lin_coord_y(dy, scale_y, sy, b1);
        const float b0 = 1.f - b1;
        const int r0 = clampi(sy, 0, sh - 1), r1 = clampi(sy + 1, 0, sh - 1);
        if (r0 != ra) {                                       // all branches are wave-uniform
            if (r0 == rb) {
#pragma unroll
                for (int i = 0; i < NPL; i++) hA[i] = hB[i];
            } else {
                hrow(r0, hA);
            }
            ra = r0;
        }
        const bool same = r1 == ra;
        if (!same && r1 != rb) {
            hrow(r1, hB);
            rb = r1;
        }
        if (live) {
            float2 o[NPL];
#pragma unroll
            for (int i = 0; i < NPL; i++) {
                const float2 t0 = hA[i], t1 = same ? hA[i] : hB[i];
                o[i].x = NSOF_MADD(t0.x, b0, t1.x * b1) * mul;
                o[i].y = NSOF_MADD(t0.y, b0, t1.y * b1) * mul;
            }
            float2* drow = D + (size_t)dy * dw + dx0;
            if (dx0 + 1 < dw && (dw & 1) == 0) {
                nsof_store_stream4(reinterpret_cast<float*>(drow), o[0].x, o[0].y, o[1].x, o[1].y);
            } else {
                drow[0] = o[0];
                if (dx0 + 1 < dw) drow[1] = o[1];
            }
        }
    }
}

template <int N>
void launch_polyexp_n(nsof_ctx* ctx, int n_img, const float* img, int W, int H, const nsof_poly_taps& taps, float* R,
                      const PolyU8* u8 = nullptr)
{
    using G = PolyGeom<N>;
    const int strips = (W + G::SW - 1) / G::SW;
    // segment the height so that the grid has >= ~2048 blocks, but keep segments >= 64 rows
    int segs = 1;
    while (segs < 64 && (long long)strips * n_img * segs < 2048 && (H / (segs * 2)) >= 64) segs *= 2;
    // a lone pair: segments down to 16 rows (each re-loads 2N+1 rows of warm-up) until there is a workgroup per CU
    while (segs < 128 && (long long)strips * n_img * segs < 256 && (H / (segs * 2)) >= 16) segs *= 2;
    int seg_rows = ((H + segs - 1) / segs + 3) / 4 * 4;
    segs = (H + seg_rows - 1) / seg_rows;
    dim3 grid(strips, segs, n_img);
#ifdef NSOF_AB
    static const bool mono = [] { const char* e = NSOF_AB_GETENV("NSOF_POLYEXP"); return e && e[0] == 'm'; }();   // A/B: single-role kernel
#endif
    if (u8)
        hipLaunchKernelGGL((k_polyexp_rs<N, false, true>), grid, dim3(512), 0, ctx->stream, img, R, W, H, seg_rows, taps, nullptr, *u8);
    else if (ctx->opt_polyexp_f32)
        hipLaunchKernelGGL((k_polyexp<N, false, true>), grid, dim3(256), 0, ctx->stream, img, R, W, H, seg_rows, taps,
                           nullptr);
#ifdef NSOF_AB   // NSOF_POLYEXP=mono: the single-role kernel with the exact arithmetic (superseded by k_polyexp_rs)
    else if (mono)
        hipLaunchKernelGGL((k_polyexp<N, false>), grid, dim3(256), 0, ctx->stream, img, R, W, H, seg_rows, taps, nullptr);
#endif
    else
        hipLaunchKernelGGL((k_polyexp_rs<N, false>), grid, dim3(512), 0, ctx->stream, img, R, W, H, seg_rows, taps, nullptr);
}

// Work-list twin: W, H are the largest level extents over the table, n_img = 2 * items.
template <int N>
void launch_polyexp_het_n(nsof_ctx* ctx, int n_img, const nsof_het_item* items, const float* img, int W, int H,
                          const nsof_poly_taps& taps, float* R, const PolyU8* u8 = nullptr)
{
    using G = PolyGeom<N>;
    const int strips = (W + G::SW - 1) / G::SW;
    int segs = 1;
    while (segs < 64 && (long long)strips * n_img * segs < 2048 && (H / (segs * 2)) >= 64) segs *= 2;
    int seg_rows = ((H + segs - 1) / segs + 3) / 4 * 4;
    segs = (H + seg_rows - 1) / seg_rows;
    dim3 grid(strips, segs, n_img);
    if (u8) hipLaunchKernelGGL((k_polyexp_rs<N, true, true>), grid, dim3(512), 0, ctx->stream, img, R, W, H, seg_rows, taps, items, *u8);
    else hipLaunchKernelGGL((k_polyexp_rs<N, true>), grid, dim3(512), 0, ctx->stream, img, R, W, H, seg_rows, taps, items);
}

}  // namespace

// =========================================================================================
// launchers
// =========================================================================================
// Levels 1..3 of a pyr_scale 0.5 pyramid in one launch (k_prep_decim3).  Returns NSOF_EUNSUPPORTED without launching
// when the frames do not decimate exactly by 8 (or are not aligned for the vector walks): the caller then runs the
// levels one by one.  out[k - 1]: level k's images, [n_img][H >> k][W >> k].
int NSOF_PYR_NAME(nsof_launch_prep_decim3)(nsof_ctx* ctx, int n_img, const uint8_t* src, ptrdiff_t row_stride, ptrdiff_t img_stride,
                                           int W, int H, const nsof_blur_taps* taps, float* const* out)
{
    const int CWL = (W & 15) == 0 ? 16 : 8;
    const bool ok = (W & 7) == 0 && (H & 7) == 0 && W >= 64 && H > 19 && (row_stride % CWL) == 0 && (img_stride % CWL) == 0 &&
                    (reinterpret_cast<uintptr_t>(src) % CWL) == 0 && taps[0].ksize == 3 && taps[1].ksize == 9 &&
                    taps[2].ksize == 19 && n_img <= 65535;
    if (!ok) return NSOF_EUNSUPPORTED;
    nsof_prof_scope ps(ctx, NSOF_K_PREP);
    Decim3 d;
    for (int k = 0; k < 3; k++) {
        d.t[k] = taps[k];
        d.out[k] = out[k];
    }
    // a wave's segment covers the same 96 source rows at every level (48 / 24 / 12 output rows: multiples of the walks'
    // unroll counts 2 / 3 / 3); few images: shorter segments, as in the one-level launcher
    int src_rows = 96;
    const long waves_x = (W / CWL + 63) / 64;
    while (src_rows > 24 && waves_x * ((H + src_rows - 1) / src_rows) * n_img < 1024) src_rows -= 24;
    d.seg_rows[0] = src_rows / 2;
    d.seg_rows[1] = src_rows / 4;
    d.seg_rows[2] = src_rows / 8;
    const int nseg = (H / 8 + d.seg_rows[2] - 1) / d.seg_rows[2];
    dim3 grid((unsigned)waves_x, (nseg + 3) / 4, n_img);
    if (CWL == 16) hipLaunchKernelGGL(k_prep_decim3<16>, grid, dim3(768), 0, ctx->stream, src, row_stride, img_stride, W, H, d);
    else hipLaunchKernelGGL(k_prep_decim3<8>, grid, dim3(768), 0, ctx->stream, src, row_stride, img_stride, W, H, d);
    NSOF_HIP(ctx, hipGetLastError());
    return NSOF_OK;
}

int NSOF_PYR_NAME(nsof_launch_prep)(nsof_ctx* ctx, int n_img, const uint8_t* src, ptrdiff_t row_stride, ptrdiff_t img_stride, int W,
                     int H, int wk, int hk, const nsof_blur_taps& taps, float* out)
{
    nsof_prof_scope ps(ctx, NSOF_K_PREP);
    if (wk == W && hk == H) {
        const bool aligned = (W & 3) == 0 && (row_stride & 3) == 0 && (img_stride & 3) == 0 &&
                             (reinterpret_cast<uintptr_t>(src) & 3) == 0 && W >= 8;
        if (taps.ksize == 3 && aligned) {
            dim3 grid((W / 4 + 63) / 64, (H + 4 * PREP0_ROWS - 1) / (4 * PREP0_ROWS), n_img);
            hipLaunchKernelGGL(k_prep_same3_vec<false>, grid, dim3(256), 0, ctx->stream, src, row_stride, img_stride, W,
                               H, taps.k[1], taps.k[2], out, nullptr);
        } else {
            dim3 grid((W + 63) / 64, (H + 3) / 4, n_img);
            hipLaunchKernelGGL(k_prep_same<false>, grid, dim3(256), 0, ctx->stream, src, row_stride, img_stride, W, H,
                               taps, out, nullptr, -1);
        }
    } else {
        const double scale_x = 1. / ((double)wk / W), scale_y = 1. / ((double)hk / H);
        const int r = taps.ksize / 2;
        const int rw_cap = ((int)ceil(PREP_TW * scale_x) + 2 * r + 8 + 3) / 4 * 4;
        const int rh_cap = (int)ceil(PREP_TH * scale_y) + 2 * r + 4;
        const size_t smem = sizeof(float) * ((size_t)rh_cap * 2 * PREP_TW + 2 * PREP_TH * 2 * PREP_TW) +
                            (size_t)rh_cap * rw_cap;
        const bool direct_ok = scale_x >= 1.0 && scale_y >= 1.0 &&
                               (taps.ksize == 3 || taps.ksize == 5);   // larger kernels: registers run out
        // exact decimation by 2 / 4 / 8 with the kernel sizes the pyr_scale 0.5 pyramid produces
        const int S = W / wk;
        static const bool force8 = NSOF_AB_GETENV("NSOF_DECIM_CW8") != nullptr;   // A/B: 8-column lanes everywhere
        const int CWL = ((W & 15) == 0 && !force8) ? 16 : 8;   // source columns per lane
        const bool decim_ok = S >= 2 && W == S * wk && H == S * hk && (W % CWL) == 0 && W >= 64 &&
                              (row_stride % CWL) == 0 && (img_stride % CWL) == 0 &&
                              (reinterpret_cast<uintptr_t>(src) % CWL) == 0 &&
                              ((S == 2 && taps.ksize == 3) || (S == 4 && taps.ksize == 9) ||
                               (S == 8 && taps.ksize == 19)) &&
                              H > taps.ksize && NSOF_AB_GETENV("NSOF_PREP_NODECIM") == nullptr;
        if (decim_ok) {
            // segments of output rows: multiples of the unroll count, ~16 source rows of warm-up amortised
            const int U = S == 2 ? 2 : 3;
            int seg_rows = S == 2 ? 32 : (S == 4 ? 24 : 15);
            seg_rows = (seg_rows + U - 1) / U * U;
            {
                // few images (one call per camera frame): shorter segments so that the launch still has ~1000 waves;
                // a segment re-runs ~KS+1 source rows of warm-up, which only matters when the GPU is full anyway
                const long waves_x = (W / CWL + 63) / 64;
                const long have = waves_x * ((hk + seg_rows - 1) / seg_rows) * n_img;
                if (have < 1024) {
                    const long want_seg = (1024 + waves_x * n_img - 1) / (waves_x * n_img);
                    const int rows = (int)std::max<long>(U, (hk / want_seg) / U * U);
                    if (rows < seg_rows) seg_rows = rows;
                }
            }
            const int nseg = (hk + seg_rows - 1) / seg_rows;
            dim3 grid((W / CWL + 63) / 64, (nseg + 3) / 4, n_img);
#define NSOF_DECIM(SS, KK, CC)                                                                                      \
    hipLaunchKernelGGL((k_prep_decim<SS, KK, CC>), grid, dim3(256), 0, ctx->stream, src, row_stride, img_stride, W, \
                       H, wk, hk, seg_rows, taps, out)
            if (CWL == 16) {
                if (S == 2) NSOF_DECIM(2, 3, 16);
                else if (S == 4) NSOF_DECIM(4, 9, 16);
                else NSOF_DECIM(8, 19, 16);
            } else {
                if (S == 2) NSOF_DECIM(2, 3, 8);
                else if (S == 4) NSOF_DECIM(4, 9, 8);
                else NSOF_DECIM(8, 19, 8);
            }
#undef NSOF_DECIM
        } else if (scale_x >= 1.0 && scale_y >= 1.0 && scale_y < taps.ksize &&
                   (taps.ksize == 3 || taps.ksize == 5 || taps.ksize == 9)) {
            // measured per 128-image launch at 1080p, pyr_scale 0.6: level 1 (3 taps) 637 -> 334 us, level 2 (5 taps) 403 ->
            // 266 us against the direct kernel (static-slot ring: 367 / 322 us with a shifting ring).  Round 4 (scalar
            // control flow, per-wave fast path, a block of loads in flight; per 256 images): 3 taps 658 -> 497 us, 5 taps
            // 519 -> 490 us, and with 9 taps (level 3: 415 x 233 outputs, 4.6 source rows per destination row) the walk
            // now beats the tiled kernel too (762 vs 813 us), which it lost to before (624 vs 415 us per 128)
            // segments of destination rows: long enough that the KS+1 rows of warm-up are a few per cent, short enough
            // that a small batch still has a few thousand waves
            int seg_rows = 32;
            const long waves_x = (wk + 63) / 64;
            while (seg_rows > 8 && waves_x * ((hk + seg_rows - 1) / seg_rows) * n_img < 2048) seg_rows /= 2;
            const int nseg = (hk + seg_rows - 1) / seg_rows;
            dim3 grid((unsigned)waves_x, (nseg + 3) / 4, n_img);
#define NSOF_PREP_WALK(KS)                                                                                          \
    hipLaunchKernelGGL((k_prep_walk<KS>), grid, dim3(256), 0, ctx->stream, src, row_stride, img_stride, W, H, wk, hk, \
                       scale_x, scale_y, seg_rows, taps, out)
            if (taps.ksize == 3) NSOF_PREP_WALK(3);
            else if (taps.ksize == 5) NSOF_PREP_WALK(5);
            else NSOF_PREP_WALK(9);
#undef NSOF_PREP_WALK
        } else if (direct_ok) {
            dim3 grid((wk + 63) / 64, (hk + 3) / 4, n_img);
#define NSOF_PREP_DIRECT(KS)                                                                                       \
    hipLaunchKernelGGL((k_prep_direct<KS, false>), grid, dim3(256), 0, ctx->stream, src, row_stride, img_stride, W, \
                       H, wk, hk, scale_x, scale_y, taps, out, nullptr)
            if (taps.ksize == 3) NSOF_PREP_DIRECT(3);
            else NSOF_PREP_DIRECT(5);
#undef NSOF_PREP_DIRECT
        } else if (taps.ksize == 19 && scale_x >= 1.0 && scale_y >= 1.0 && NSOF_AB_GETENV("NSOF_PREP_TILED") == nullptr) {
            // measured at 1080p x 64 frames: 19 taps 459 -> 244 us; 9 taps is still faster tiled (231 vs 254 us)
            int rc = nsof_ws_reserve(ctx, &ctx->tmp, &ctx->tmp_bytes, (size_t)n_img * H * 2 * wk * sizeof(float));
            if (rc) return rc;
            float* HA = static_cast<float*>(ctx->tmp);
            dim3 ga((2 * wk + 63) / 64, (H + 4 * PREPA_ROWS - 1) / (4 * PREPA_ROWS), n_img);
            dim3 gb((wk + 63) / 64, (hk + 3) / 4, n_img);
            hipLaunchKernelGGL(k_prep_rows<19>, ga, dim3(256), 0, ctx->stream, src, row_stride, img_stride, W, H, wk,
                               scale_x, taps, HA);
            hipLaunchKernelGGL(k_prep_cols<19>, gb, dim3(256), 0, ctx->stream, HA, W, H, wk, hk, scale_x, scale_y, taps,
                               out);
        } else if (smem <= 60 * 1024 && scale_x >= 1.0 && scale_y >= 1.0) {
            dim3 grid((wk + PREP_TW - 1) / PREP_TW, (hk + PREP_TH - 1) / PREP_TH, n_img);
#define NSOF_PREP_TILED(KS)                                                                                        \
    hipLaunchKernelGGL((k_prep_tiled<KS, false>), grid, dim3(256), smem, ctx->stream, src, row_stride, img_stride, \
                       W, H, wk, hk, scale_x, scale_y, rw_cap, rh_cap, taps, out, nullptr)
            switch (taps.ksize) {
                case 9: NSOF_PREP_TILED(9); break;
                case 19: NSOF_PREP_TILED(19); break;
                default: NSOF_PREP_TILED(0); break;
            }
#undef NSOF_PREP_TILED
        } else {
            dim3 grid((wk + 63) / 64, (hk + 3) / 4, n_img);
            hipLaunchKernelGGL(k_prep_naive<false>, grid, dim3(256), 0, ctx->stream, src, row_stride, img_stride, W, H,
                               wk, hk, scale_x, scale_y, taps, out, nullptr);
        }
    }
    NSOF_HIP(ctx, hipGetLastError());
    return NSOF_OK;
}

#ifndef NSOF_PYR_FMA
// The expansion of the full-resolution level straight from the 8-bit frames (k_polyexp_rs<.., U8>): images [0, nsplit)
// at src0 + z * img_stride, the rest at src1; k0 / k1 = centre / side tap of the level's 3-tap smoothing.
int nsof_launch_polyexp_u8(nsof_ctx* ctx, int n_img, const uint8_t* src0, const uint8_t* src1, int nsplit, ptrdiff_t row_stride,
                           ptrdiff_t img_stride, int W, int H, const nsof_poly_taps& taps, float k0, float k1, float* R)
{
    nsof_prof_scope ps(ctx, NSOF_K_POLYEXP);
    const PolyU8 u8{src0, src1, (long long)row_stride, (long long)img_stride, nsplit, k0, k1};
    switch (taps.n) {
#define NSOF_PU(NN) case NN: launch_polyexp_n<NN>(ctx, n_img, nullptr, W, H, taps, R, &u8); break
        NSOF_PU(1); NSOF_PU(2); NSOF_PU(3); NSOF_PU(4); NSOF_PU(5); NSOF_PU(6); NSOF_PU(7); NSOF_PU(8); NSOF_PU(9); NSOF_PU(10);
#undef NSOF_PU
        default: return nsof_set_error(ctx, NSOF_EUNSUPPORTED, "poly_n=%d outside 1..%d", taps.n, NSOF_MAX_POLY_N);
    }
    NSOF_HIP(ctx, hipGetLastError());
    return NSOF_OK;
}

int nsof_launch_polyexp(nsof_ctx* ctx, int n_img, const float* img, int W, int H, const nsof_poly_taps& taps, float* R)
{
    nsof_prof_scope ps(ctx, NSOF_K_POLYEXP);
    switch (taps.n) {
        case 1: launch_polyexp_n<1>(ctx, n_img, img, W, H, taps, R); break;
        case 2: launch_polyexp_n<2>(ctx, n_img, img, W, H, taps, R); break;
        case 3: launch_polyexp_n<3>(ctx, n_img, img, W, H, taps, R); break;
        case 4: launch_polyexp_n<4>(ctx, n_img, img, W, H, taps, R); break;
        case 5: launch_polyexp_n<5>(ctx, n_img, img, W, H, taps, R); break;
        case 6: launch_polyexp_n<6>(ctx, n_img, img, W, H, taps, R); break;
        case 7: launch_polyexp_n<7>(ctx, n_img, img, W, H, taps, R); break;
        case 8: launch_polyexp_n<8>(ctx, n_img, img, W, H, taps, R); break;
        case 9: launch_polyexp_n<9>(ctx, n_img, img, W, H, taps, R); break;
        case 10: launch_polyexp_n<10>(ctx, n_img, img, W, H, taps, R); break;
        default: return nsof_set_error(ctx, NSOF_EUNSUPPORTED, "poly_n=%d outside 1..%d", taps.n, NSOF_MAX_POLY_N);
    }
    NSOF_HIP(ctx, hipGetLastError());
    return NSOF_OK;
}

int nsof_launch_update_matrices(nsof_ctx* ctx, int n_pairs, const float* R0, const float* R1, size_t pair_stride,
                                const float* flow, int W, int H, float* M)
{
    nsof_prof_scope ps(ctx, NSOF_K_UPDMAT);
    dim3 grid((W + 63) / 64, (H + 3) / 4, n_pairs);
    hipLaunchKernelGGL(k_update_matrices, grid, dim3(256), 0, ctx->stream, R0, R1, pair_stride, flow, W, H, M);
    NSOF_HIP(ctx, hipGetLastError());
    return NSOF_OK;
}

int nsof_launch_blur_solve(nsof_ctx* ctx, int n_pairs, const float* M, int W, int H, int winsize, float* flow)
{
    const int m = winsize / 2;
    if (m > 96) return nsof_set_error(ctx, NSOF_EUNSUPPORTED, "winsize=%d too large (max 193)", winsize);
    nsof_prof_scope ps(ctx, NSOF_K_BLUR);
    const int SW = (256 - 2 * m) & ~3;
    dim3 grid((W + SW - 1) / SW, 1, n_pairs);
    hipLaunchKernelGGL(k_blur_solve, grid, dim3(256), 0, ctx->stream, M, W, H, m, winsize, flow);
    NSOF_HIP(ctx, hipGetLastError());
    return NSOF_OK;
}

int nsof_launch_blur_solve_exact(nsof_ctx* ctx, int n_pairs, const float* M, int W, int H, int winsize, double* VT,
                                 float* flow)
{
    const int m = winsize / 2;
    nsof_prof_scope ps(ctx, NSOF_K_BLUR);
    hipLaunchKernelGGL(k_blur_colsum, dim3((W + 255) / 256, 1, n_pairs), dim3(256), 0, ctx->stream, M, W, H, m, VT);
    hipLaunchKernelGGL(k_blur_rowsolve, dim3((H + 63) / 64, 1, n_pairs), dim3(64), 0, ctx->stream, VT, W, H, m, winsize,
                       flow);
    NSOF_HIP(ctx, hipGetLastError());
    return NSOF_OK;
}

#endif  // !NSOF_PYR_FMA

int NSOF_PYR_NAME(nsof_launch_flow_upsample)(nsof_ctx* ctx, int n_pairs, const float* src, int sw, int sh, float* dst, int dw, int dh,
                              float mul)
{
    nsof_prof_scope ps(ctx, NSOF_K_UPSAMPLE);
    const double scale_x = 1. / ((double)dw / sw), scale_y = 1. / ((double)dh / sh);
    if (dw >= sw && dh >= sh && sw >= 1 && sh >= 1 && dw >= 256 && NSOF_AB_GETENV("NSOF_UPSAMPLE_2X2") == nullptr) {
        // upsampling: source steps of 0 or 1 between neighbours; rows wide enough for a lane per 2 columns
        dim3 g(((dw + 1) / 2 + 255) / 256, (dh + UPW_SEG - 1) / UPW_SEG, n_pairs);
        hipLaunchKernelGGL(k_flow_upsample_walk, g, dim3(256), 0, ctx->stream, src, sw, sh, dst, dw, dh, scale_x,
                           scale_y, mul);
        NSOF_HIP(ctx, hipGetLastError());
        return NSOF_OK;
    }
    if (dw >= sw && dh >= sh && sw >= 1 && sh >= 1) {   // upsampling: source steps of 0 or 1 between neighbours
        dim3 g2(((dw + 1) / 2 + 63) / 64, ((dh + 1) / 2 + 3) / 4, n_pairs);
        hipLaunchKernelGGL(k_flow_upsample2x2<false>, g2, dim3(256), 0, ctx->stream, src, sw, sh, dst, dw, dh, scale_x,
                           scale_y, mul, nullptr);
        NSOF_HIP(ctx, hipGetLastError());
        return NSOF_OK;
    }
    dim3 grid((dw + 63) / 64, (dh + 3) / 4, n_pairs);
    hipLaunchKernelGGL(k_flow_upsample, grid, dim3(256), 0, ctx->stream, src, sw, sh, dst, dw, dh, scale_x, scale_y,
                       mul);
    NSOF_HIP(ctx, hipGetLastError());
    return NSOF_OK;
}

// =========================================================================================
// work-list (shape-heterogeneous) launchers: one launch per stage and level over a device table
// =========================================================================================
int NSOF_PYR_NAME(nsof_launch_prep_het)(nsof_ctx* ctx, int n_items, const nsof_het_item* d_items, const nsof_het_item* h_items,
                         bool level0, const nsof_blur_taps& taps, float* I)
{
    nsof_prof_scope ps(ctx, NSOF_K_PREP);
    int max_wk = 0, max_hk = 0, n_vec = 0;
    double max_sx = 1, max_sy = 1;
    for (int i = 0; i < n_items; i++) {
        const nsof_het_item& it = h_items[i];
        max_wk = it.wk > max_wk ? it.wk : max_wk;
        max_hk = it.hk > max_hk ? it.hk : max_hk;
        n_vec += (it.flags & NSOF_HET_VEC0) ? 1 : 0;
        const double sx = 1. / ((double)it.wk / it.W), sy = 1. / ((double)it.hk / it.H);
        max_sx = sx > max_sx ? sx : max_sx;
        max_sy = sy > max_sy ? sy : max_sy;
    }
    const int nz = 2 * n_items;
    if (level0) {   // same-size level: 3 taps; aligned items take the vector kernel, the others the generic one
        if (taps.ksize == 3 && n_vec > 0) {
            dim3 grid((max_wk / 4 + 63) / 64, (max_hk + 4 * PREP0_ROWS - 1) / (4 * PREP0_ROWS), nz);
            hipLaunchKernelGGL(k_prep_same3_vec<true>, grid, dim3(256), 0, ctx->stream, nullptr, 0, 0, 0, 0, taps.k[1],
                               taps.k[2], I, d_items);
        }
        if (taps.ksize != 3 || n_vec < n_items) {
            dim3 grid((max_wk + 63) / 64, (max_hk + 3) / 4, nz);
            hipLaunchKernelGGL(k_prep_same<true>, grid, dim3(256), 0, ctx->stream, nullptr, 0, 0, 0, 0, taps, I, d_items,
                               taps.ksize == 3 ? 0 : -1);
        }
    } else {
        const int r = taps.ksize / 2;
        const int rw_cap = ((int)ceil(PREP_TW * max_sx) + 2 * r + 8 + 3) / 4 * 4;
        const int rh_cap = (int)ceil(PREP_TH * max_sy) + 2 * r + 4;
        const size_t smem = sizeof(float) * ((size_t)rh_cap * 2 * PREP_TW + 2 * PREP_TH * 2 * PREP_TW) +
                            (size_t)rh_cap * rw_cap;
        if (taps.ksize == 3 || taps.ksize == 5) {
            dim3 grid((max_wk + 63) / 64, (max_hk + 3) / 4, nz);
            if (taps.ksize == 3)
                hipLaunchKernelGGL((k_prep_direct<3, true>), grid, dim3(256), 0, ctx->stream, nullptr, 0, 0, 0, 0, 0, 0, 1.,
                                   1., taps, I, d_items);
            else
                hipLaunchKernelGGL((k_prep_direct<5, true>), grid, dim3(256), 0, ctx->stream, nullptr, 0, 0, 0, 0, 0, 0, 1.,
                                   1., taps, I, d_items);
        } else if (smem <= 60 * 1024) {
            dim3 grid((max_wk + PREP_TW - 1) / PREP_TW, (max_hk + PREP_TH - 1) / PREP_TH, nz);
#define NSOF_PREP_TILED_HET(KS)                                                                                       \
    hipLaunchKernelGGL((k_prep_tiled<KS, true>), grid, dim3(256), smem, ctx->stream, nullptr, 0, 0, 0, 0, 0, 0, 1., 1., \
                       rw_cap, rh_cap, taps, I, d_items)
            switch (taps.ksize) {
                case 9: NSOF_PREP_TILED_HET(9); break;
                case 19: NSOF_PREP_TILED_HET(19); break;
                default: NSOF_PREP_TILED_HET(0); break;
            }
#undef NSOF_PREP_TILED_HET
        } else {
            dim3 grid((max_wk + 63) / 64, (max_hk + 3) / 4, nz);
            hipLaunchKernelGGL(k_prep_naive<true>, grid, dim3(256), 0, ctx->stream, nullptr, 0, 0, 0, 0, 0, 0, 1., 1.,
                               taps, I, d_items);
        }
    }
    NSOF_HIP(ctx, hipGetLastError());
    return NSOF_OK;
}

#ifndef NSOF_PYR_FMA
// blur3: non-null at the full-resolution level = form the level image from the items' own frames (k0, k1 = centre / side
// tap); I is not read then.
int nsof_launch_polyexp_het(nsof_ctx* ctx, int n_items, const nsof_het_item* d_items, int max_w, int max_h,
                            const nsof_poly_taps& taps, const float* I, float* R, const float* blur3)
{
    nsof_prof_scope ps(ctx, NSOF_K_POLYEXP);
    const int nz = 2 * n_items;
    PolyU8 u8v{};
    if (blur3) { u8v.k0 = blur3[0]; u8v.k1 = blur3[1]; }
    const PolyU8* u8 = blur3 ? &u8v : nullptr;
    switch (taps.n) {
#define NSOF_PH(NN) case NN: launch_polyexp_het_n<NN>(ctx, nz, d_items, I, max_w, max_h, taps, R, u8); break
        NSOF_PH(1); NSOF_PH(2); NSOF_PH(3); NSOF_PH(4); NSOF_PH(5); NSOF_PH(6); NSOF_PH(7); NSOF_PH(8); NSOF_PH(9); NSOF_PH(10);
#undef NSOF_PH
        default: return nsof_set_error(ctx, NSOF_EUNSUPPORTED, "poly_n=%d outside 1..%d", taps.n, NSOF_MAX_POLY_N);
    }
    NSOF_HIP(ctx, hipGetLastError());
    return NSOF_OK;
}

#endif  // !NSOF_PYR_FMA

int NSOF_PYR_NAME(nsof_launch_flow_upsample_het)(nsof_ctx* ctx, int n_items, const nsof_het_item* d_items, int max_w, int max_h,
                                  const float* src, float* dst, float mul)
{
    nsof_prof_scope ps(ctx, NSOF_K_UPSAMPLE);
    dim3 g2(((max_w + 1) / 2 + 63) / 64, ((max_h + 1) / 2 + 3) / 4, n_items);
    hipLaunchKernelGGL(k_flow_upsample2x2<true>, g2, dim3(256), 0, ctx->stream, src, 0, 0, dst, 0, 0, 1., 1., mul,
                       d_items);
    NSOF_HIP(ctx, hipGetLastError());
    return NSOF_OK;
}
