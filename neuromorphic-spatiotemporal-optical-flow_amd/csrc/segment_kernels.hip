// Motion-segmentation head on the flow field (reference optical_flow_seg.py:253-357, 503-537):
//   mask = |flow| > SEG_TH ; 5 x (dilate, erode) with a 10x10 elliptical element ; 0/255.
//
// The masks are two-valued, so the head works on BIT-PACKED rows (one 32-bit word = 32 pixels):
//   k_mag_pack    flow [h][w][2] f32 -> bits, one wave ballot per 64 pixels.  HBM-bound: 8 B/px in, 1/8 B/px out.
//   k_u8_pack     any u8 image (non-zero = set) -> bits, for the dilate/erode mirror.
//   k_morph_bits  ALL passes of a dilate/erode chain in one launch.  A workgroup owns an output tile plus the halo
//                 the whole chain needs (passes * element reach), holds it in LDS, and per pass does
//                   H step: for every distinct row pattern of the element, OR of the funnel-shifted words
//                           (v_alignbit: one instruction per element column, 32 pixels at a time)
//                   V step: OR over the element rows of the matching H array
//                 erode = complement . dilate . complement with the same offsets (cv2 does not reflect the element).
//                 Pixels outside the image never take part (cv2's default border for morphology), which is the
//                 `inside` mask applied after every pass.  The bit image is 1/64 of the flow's size, so the halo
//                 re-reads are free and the head as a whole stays bound by the one read of the flow.
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "nsof_internal.h"

namespace {

constexpr int MAX_K = 32;         // element up to 32x32 (one 32-bit row pattern)
constexpr int MAX_DISTINCT = 16;
constexpr int TILE_H = 32;        // output rows per workgroup
constexpr int TILE_WORDS = 12;    // output words per workgroup (384 px)
constexpr int HALO_WORDS = 2;     // 64 px each side
constexpr int TW = TILE_WORDS + 2 * HALO_WORDS;   // 16 words per LDS row
constexpr int MORPH_THREADS = 1024;
constexpr int MAX_ROWS = 256;     // LDS rows per tile
static_assert(TW == 16, "index arithmetic below uses shifts by 4");

struct MorphElem {
    uint32_t extra[MAX_DISTINCT];    // pattern d = pattern base[d] | extra[d]   (bit j = element column j)
    int8_t base[MAX_DISTINCT];       // an earlier pattern that is a subset of d, or -1
    uint8_t row_pattern[MAX_K];      // pattern index of element row i, 0xff = empty row
    int n_patterns, kw, kh, ax, ay;
};

__global__ __launch_bounds__(256) void k_mag_pack(const float* __restrict__ flow, ptrdiff_t fstride, int w, int h,
                                                  double thresh, uint32_t* __restrict__ bits, int wp)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k64 = blockIdx.x * 4 + wave;   // 64-pixel group of the row
    if (k64 * 2 >= wp) return;
    const int x = k64 * 64 + lane;
    const int y0 = blockIdx.y * 8;
    float2 v[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int y = y0 + r;
        v[r] = (x < w && y < h) ? *(const float2*)(flow + (ptrdiff_t)y * fstride + 2 * x) : make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int y = y0 + r;
        const double a = v[r].x, b = v[r].y;
        const bool set = x < w && y < h && sqrt(a * a + b * b) > thresh;   // cartToPolar on float64, then `mag > th`
        const unsigned long long m = __ballot(set);
        if (lane == 0 && y < h) *(unsigned long long*)(bits + (size_t)y * wp + 2 * k64) = m;
    }
}

__global__ __launch_bounds__(256) void k_u8_pack(const uint8_t* __restrict__ src, ptrdiff_t sstride, int w, int h,
                                                 uint32_t* __restrict__ bits, int wp)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k64 = blockIdx.x * 4 + wave;
    if (k64 * 2 >= wp) return;
    const int x = k64 * 64 + lane;
    const int y0 = blockIdx.y * 8;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int y = y0 + r;
        const bool set = x < w && y < h && src[(ptrdiff_t)y * sstride + x] != 0;
        const unsigned long long m = __ballot(set);
        if (lane == 0 && y < h) *(unsigned long long*)(bits + (size_t)y * wp + 2 * k64) = m;
    }
}

// Bits of word gk of row gy that lie inside the image.
__device__ __forceinline__ uint32_t inside_bits(int gy, int gk, int w, int h)
{
    if (gy < 0 || gy >= h || gk < 0) return 0u;
    const int left = w - gk * 32;   // pixels of the row from this word on
    return left >= 32 ? 0xffffffffu : (left <= 0 ? 0u : (1u << left) - 1u);
}

// Static recursion over the element description: every access to `el` has a compile-time index, so the whole
// description sits in scalar registers (indexing a kernel argument dynamically costs a scalar memory load per tap,
// which made the first version of this kernel 4x slower).
template <int D>
__device__ __forceinline__ void h_patterns(const MorphElem& el, uint32_t* H, int n, int i, uint32_t lo, uint32_t mid,
                                           uint32_t hi)
{
    if constexpr (D < MAX_DISTINCT) {
        if (D >= el.n_patterns) return;
        uint32_t acc = el.base[D] >= 0 ? H[el.base[D] * n + i] : 0u;
        for (uint32_t m = el.extra[D]; m; m &= m - 1) {   // constant trip count (unrolled) when FIXED10
            const int c = __builtin_ctz(m) - el.ax;    // out bit b takes in bit b + c
            acc |= c == 0 ? mid
                          : (c > 0 ? __builtin_amdgcn_alignbit(hi, mid, (unsigned)c)
                                   : __builtin_amdgcn_alignbit(mid, lo, (unsigned)(32 + c)));
        }
        H[D * n + i] = acc;
        h_patterns<D + 1>(el, H, n, i, lo, mid, hi);
    }
}

template <int E>
__device__ __forceinline__ uint32_t v_rows(const MorphElem& el, const uint32_t* H, int n, int r, int k, int rows)
{
    if constexpr (E < MAX_K) {
        if (E >= el.kh) return 0u;
        const int d = el.row_pattern[E];
        const int rr = r + E - el.ay;
        const uint32_t v = (d != 0xff && rr >= 0 && rr < rows) ? H[d * n + rr * TW + k] : 0u;
        return v | v_rows<E + 1>(el, H, n, r, k, rows);
    } else {
        return 0u;
    }
}

// The element of the reference (10x10 ellipse, centre anchor) as a compile-time constant: with it the H and V steps
// unroll into straight-line code (10 funnel shifts, 10 LDS reads per word); the run-time description costs scalar
// control flow per tap and is ~6x slower per pass.
constexpr MorphElem ellipse10()
{
    MorphElem e{};
    e.extra[0] = 0x020u; e.extra[1] = 0x1DCu; e.extra[2] = 0x202u; e.extra[3] = 0x001u;   // rows of 1, 7, 9, 10 pixels
    e.base[0] = -1; e.base[1] = 0; e.base[2] = 1; e.base[3] = 2;
    constexpr uint8_t rp[10] = {0, 1, 2, 3, 3, 3, 3, 3, 2, 1};
    for (int i = 0; i < 10; i++) e.row_pattern[i] = rp[i];
    e.n_patterns = 4; e.kw = 10; e.kh = 10; e.ax = 5; e.ay = 5;
    return e;
}

// ops: bit p = 1 -> pass p is a dilate, 0 -> erode.  Output: out_u8 (0/255) when non-null, else out_bits.
// FIXED10: ignore el_arg and use ellipse10().
template <bool FIXED10>
__global__ __launch_bounds__(MORPH_THREADS) void k_morph_bits(const uint32_t* __restrict__ in_bits, int wp, int w,
                                                              int h, const MorphElem el_arg, int n_pass, unsigned ops,
                                                              int top, int rows, uint32_t* __restrict__ out_bits,
                                                              uint8_t* __restrict__ out_u8, ptrdiff_t ostride)
{
    constexpr MorphElem fixed = ellipse10();
    const MorphElem& el = FIXED10 ? fixed : el_arg;
    extern __shared__ uint32_t lds[];
    uint32_t* cur = lds;                       // [rows][TW]
    uint32_t* H = lds + (size_t)rows * TW;     // [n_patterns][rows][TW]
    const int tid = threadIdx.x;
    const int gk0 = blockIdx.x * TILE_WORDS - HALO_WORDS;
    const int gy0 = blockIdx.y * TILE_H - top;
    const int n = rows * TW;
    const int k = tid & (TW - 1);              // 1024 % TW == 0: a thread keeps its word column
    const int gk = gk0 + k;

    for (int i = tid; i < n; i += MORPH_THREADS) {
        const int gy = gy0 + (i >> 4);
        cur[i] = (gy >= 0 && gy < h && gk >= 0 && gk < wp) ? in_bits[(size_t)gy * wp + gk] : 0u;
    }
    __syncthreads();

    for (int p = 0; p < n_pass; p++) {
        const bool dil = (ops >> p) & 1u;
        // H step: per distinct row pattern, OR of the funnel-shifted words (one v_alignbit per element column)
        for (int i = tid; i < n; i += MORPH_THREADS) {
            const int gy = gy0 + (i >> 4);
            uint32_t lo = k > 0 ? cur[i - 1] : 0u, mid = cur[i], hi = k + 1 < TW ? cur[i + 1] : 0u;
            if (!dil) {   // complement inside the image; outside stays 0 (= "does not take part" in a minimum)
                lo = k > 0 ? ~lo & inside_bits(gy, gk - 1, w, h) : 0u;
                mid = ~mid & inside_bits(gy, gk, w, h);
                hi = k + 1 < TW ? ~hi & inside_bits(gy, gk + 1, w, h) : 0u;
            }
            h_patterns<0>(el, H, n, i, lo, mid, hi);
        }
        __syncthreads();
        // V step: OR over the element rows of the matching H array
        for (int i = tid; i < n; i += MORPH_THREADS) {
            const int r = i >> 4;
            const uint32_t acc = v_rows<0>(el, H, n, r, k, rows);
            cur[i] = (dil ? acc : ~acc) & inside_bits(gy0 + r, gk, w, h);
        }
        __syncthreads();
    }

    // write the tile's own rows/words
    if (out_u8) {
        for (int i = tid; i < TILE_H * TILE_WORDS * 8; i += MORPH_THREADS) {   // 4 pixels per item
            const int r = i / (TILE_WORDS * 8), q = i - r * (TILE_WORDS * 8);
            const int gy = blockIdx.y * TILE_H + r, gx = blockIdx.x * TILE_WORDS * 32 + q * 4;
            if (gy >= h || gx >= w) continue;
            const uint32_t word = cur[(r + top) * TW + HALO_WORDS + (q >> 3)];
            const uint32_t nib = (word >> ((q & 7) * 4)) & 0xfu;
            uint8_t* o = out_u8 + (ptrdiff_t)gy * ostride + gx;
            if (gx + 4 <= w && (((uintptr_t)o) & 3) == 0) {
                *(uint32_t*)o = ((nib & 1u) ? 0xffu : 0u) | ((nib & 2u) ? 0xff00u : 0u) | ((nib & 4u) ? 0xff0000u : 0u) |
                                ((nib & 8u) ? 0xff000000u : 0u);
            } else {
                for (int b = 0; b < 4 && gx + b < w; b++) o[b] = (nib >> b) & 1u ? 255 : 0;
            }
        }
    } else {
        for (int i = tid; i < TILE_H * TILE_WORDS; i += MORPH_THREADS) {
            const int r = i / TILE_WORDS, kk = i - r * TILE_WORDS;
            const int gy = blockIdx.y * TILE_H + r, gk = blockIdx.x * TILE_WORDS + kk;
            if (gy < h && gk < wp) out_bits[(size_t)gy * wp + gk] = cur[(r + top) * TW + HALO_WORDS + kk];
        }
    }
}

int build_elem(nsof_ctx* ctx, const uint8_t* elem, int kw, int kh, int ax, int ay, MorphElem* out)
{
    if (kw < 1 || kh < 1 || kw > MAX_K || kh > MAX_K)
        return nsof_set_error(ctx, NSOF_EUNSUPPORTED, "structuring element %dx%d: sizes 1..%d supported", kw, kh, MAX_K);
    if (ax < 0) ax = kw / 2;
    if (ay < 0) ay = kh / 2;
    if (ax >= kw || ay >= kh) return nsof_set_error(ctx, NSOF_EINVAL, "anchor (%d,%d) outside the element", ax, ay);
    MorphElem e{};
    e.kw = kw; e.kh = kh; e.ax = ax; e.ay = ay;
    uint32_t rowpat[MAX_K], pats[MAX_DISTINCT];
    int np = 0;
    for (int i = 0; i < kh; i++) {
        uint32_t pat = 0;
        for (int j = 0; j < kw; j++)
            if (elem[i * kw + j]) pat |= 1u << j;
        rowpat[i] = pat;
        if (!pat) continue;
        int d = 0;
        while (d < np && pats[d] != pat) d++;
        if (d == np) {
            if (np == MAX_DISTINCT)
                return nsof_set_error(ctx, NSOF_EUNSUPPORTED, "more than %d distinct element rows", MAX_DISTINCT);
            pats[np++] = pat;
        }
    }
    // order by population so that a pattern can build on an earlier subset (the rows of an ellipse nest)
    for (int a = 1; a < np; a++)
        for (int b = a; b > 0 && __builtin_popcount(pats[b]) < __builtin_popcount(pats[b - 1]); b--) {
            uint32_t t = pats[b]; pats[b] = pats[b - 1]; pats[b - 1] = t;
        }
    for (int d = 0; d < np; d++) {
        int best = -1;
        for (int c = 0; c < d; c++)
            if ((pats[c] & ~pats[d]) == 0 && (best < 0 || __builtin_popcount(pats[c]) > __builtin_popcount(pats[best])))
                best = c;
        e.base[d] = (int8_t)best;
        e.extra[d] = best >= 0 ? pats[d] & ~pats[best] : pats[d];
    }
    e.n_patterns = np;
    for (int i = 0; i < kh; i++) {
        int d = 0xff;
        if (rowpat[i])
            for (d = 0; pats[d] != rowpat[i]; d++) {}
        e.row_pattern[i] = (uint8_t)d;
    }
    *out = e;
    return NSOF_OK;
}

inline int words_per_row(int w) { return 2 * ((w + 63) / 64); }

// Runs the chain `ops` (n_pass passes) on bits; result to out_u8.  scratch: second bit image for multi-chunk chains.
int run_chain(nsof_ctx* ctx, uint32_t* bits, uint32_t* scratch, int w, int h, const MorphElem& el, int n_pass,
              unsigned ops, uint8_t* out_u8, ptrdiff_t ostride)
{
    const int wp = words_per_row(w);
    const int reach_x = el.ax > el.kw - 1 - el.ax ? el.ax : el.kw - 1 - el.ax;
    const int up = el.ay, down = el.kh - 1 - el.ay;
    dim3 grid((wp + TILE_WORDS - 1) / TILE_WORDS, (h + TILE_H - 1) / TILE_H);
    int done = 0;
    uint32_t* src = bits;
    uint32_t* dst = scratch;
    do {
        int chunk = n_pass - done;
        if (reach_x > 0 && chunk > (HALO_WORDS * 32) / reach_x) chunk = (HALO_WORDS * 32) / reach_x;
        size_t smem;
        int rows;
        for (;; chunk--) {   // LDS budget: (1 + patterns) arrays of rows x TW words
            rows = TILE_H + chunk * (up + down);
            smem = (size_t)(1 + (el.n_patterns ? el.n_patterns : 1)) * rows * TW * 4;
            if ((smem <= 144 * 1024 && rows <= MAX_ROWS) || chunk <= 1) break;
        }
        if (chunk < 1 || smem > 160 * 1024 || rows > MAX_ROWS)
            return nsof_set_error(ctx, NSOF_EUNSUPPORTED, "structuring element too tall for the LDS tile");
        const bool last = done + chunk >= n_pass;
        constexpr MorphElem e10 = ellipse10();
        const bool fixed10 = memcmp(&el, &e10, sizeof(MorphElem)) == 0 && NSOF_AB_GETENV("NSOF_MORPH_GENERIC") == nullptr;
        auto kern = fixed10 ? k_morph_bits<true> : k_morph_bits<false>;
        if (smem > 64 * 1024)
            NSOF_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        {
            nsof_prof_scope ps(ctx, NSOF_K_MORPH);
            hipLaunchKernelGGL(kern, grid, dim3(MORPH_THREADS), smem, ctx->stream, src, wp, w, h, el, chunk,
                               ops >> done, chunk * up, rows, last ? nullptr : dst, last ? out_u8 : nullptr, ostride);
        }
        NSOF_HIP(ctx, hipGetLastError());
        done += chunk;
        uint32_t* t = src; src = dst; dst = t;
    } while (done < n_pass);
    return NSOF_OK;
}

int reserve_bits(nsof_ctx* ctx, int w, int h, uint32_t** a, uint32_t** b)
{
    const size_t one = ((size_t)words_per_row(w) * h * 4 + 255) & ~(size_t)255;
    int rc = nsof_ws_reserve(ctx, &ctx->tmp, &ctx->tmp_bytes, 2 * one);
    if (rc) return rc;
    *a = (uint32_t*)ctx->tmp;
    *b = (uint32_t*)((char*)ctx->tmp + one);
    return NSOF_OK;
}

}  // namespace

// cv::getStructuringElement for the shapes the reference uses (ellipse) and the trivial ones.  Host arithmetic only.
extern "C" int nsof_structuring_element(int shape, int kw, int kh, uint8_t* out)
{
    if (!out || kw < 1 || kh < 1 || shape < 0 || shape > 2) return NSOF_EINVAL;
    const int r = kh / 2, c = kw / 2;
    const double inv_r2 = r ? 1.0 / ((double)r * r) : 0.0;
    for (int i = 0; i < kh; i++) {
        int j1 = 0, j2 = 0;
        if (shape == NSOF_MORPH_RECT || (shape == NSOF_MORPH_CROSS && i == r)) {
            j2 = kw;
        } else if (shape == NSOF_MORPH_CROSS) {
            j1 = c;
            j2 = c + 1;
        } else {
            const int dy = i - r;
            if (abs(dy) <= r) {
                const int dx = (int)lrint(c * sqrt((r * r - dy * dy) * inv_r2));
                j1 = c - dx > 0 ? c - dx : 0;
                j2 = c + dx + 1 < kw ? c + dx + 1 : kw;
            }
        }
        for (int j = 0; j < kw; j++) out[i * kw + j] = (uint8_t)(j >= j1 && j < j2);
    }
    return NSOF_OK;
}

extern "C" int nsof_morph_binary_u8_dev(nsof_ctx* ctx, int op, const uint8_t* d_src, ptrdiff_t src_stride, int width,
                                        int height, const uint8_t* elem, int kw, int kh, int ax, int ay,
                                        int iterations, uint8_t* d_dst, ptrdiff_t dst_stride)
{
    if (!ctx) return NSOF_EINVAL;
    if (!d_src || !d_dst || !elem) return nsof_set_error(ctx, NSOF_EINVAL, "null pointer");
    if (op != NSOF_MORPH_ERODE && op != NSOF_MORPH_DILATE) return nsof_set_error(ctx, NSOF_EINVAL, "op must be 0 or 1");
    if (width < 1 || height < 1) return nsof_set_error(ctx, NSOF_ESHAPE, "empty image");
    if (iterations < 1 || iterations > 32) return nsof_set_error(ctx, NSOF_EINVAL, "iterations must be 1..32");
    if (src_stride < width || dst_stride < width) return nsof_set_error(ctx, NSOF_EINVAL, "stride < width");
    MorphElem el;
    int rc = build_elem(ctx, elem, kw, kh, ax, ay, &el);
    if (rc) return rc;
    NSOF_HIP(ctx, hipSetDevice(ctx->device));
    uint32_t *a, *b;
    if ((rc = reserve_bits(ctx, width, height, &a, &b))) return rc;
    const int wp = words_per_row(width);
    {
        nsof_prof_scope ps(ctx, NSOF_K_SEGMENT);
        hipLaunchKernelGGL(k_u8_pack, dim3((wp / 2 + 3) / 4, (height + 7) / 8), dim3(256), 0, ctx->stream, d_src,
                           src_stride, width, height, a, wp);
    }
    NSOF_HIP(ctx, hipGetLastError());
    return run_chain(ctx, a, b, width, height, el, iterations, op == NSOF_MORPH_DILATE ? 0xffffffffu : 0u, d_dst,
                     dst_stride);
}

extern "C" int nsof_motion_mask_dev(nsof_ctx* ctx, const float* d_flow, ptrdiff_t flow_stride_floats, int width,
                                    int height, double thresh, int ksize, int iterations, uint8_t* d_mask,
                                    ptrdiff_t mask_stride)
{
    if (!ctx) return NSOF_EINVAL;
    if (!d_flow || !d_mask) return nsof_set_error(ctx, NSOF_EINVAL, "null pointer");
    if (width < 1 || height < 1) return nsof_set_error(ctx, NSOF_ESHAPE, "empty flow field");
    if (iterations < 0 || iterations > 16) return nsof_set_error(ctx, NSOF_EINVAL, "iterations must be 0..16");
    if (flow_stride_floats < 2 * (ptrdiff_t)width || (flow_stride_floats & 1) || mask_stride < width)
        return nsof_set_error(ctx, NSOF_EINVAL, "bad stride");
    if (ksize < 1 || ksize > MAX_K) return nsof_set_error(ctx, NSOF_EUNSUPPORTED, "ksize 1..%d supported", MAX_K);
    uint8_t elem[MAX_K * MAX_K];
    nsof_structuring_element(NSOF_MORPH_ELLIPSE, ksize, ksize, elem);
    MorphElem el;
    int rc = build_elem(ctx, elem, ksize, ksize, -1, -1, &el);
    if (rc) return rc;
    NSOF_HIP(ctx, hipSetDevice(ctx->device));
    uint32_t *a, *b;
    if ((rc = reserve_bits(ctx, width, height, &a, &b))) return rc;
    const int wp = words_per_row(width);
    {
        nsof_prof_scope ps(ctx, NSOF_K_SEGMENT);
        hipLaunchKernelGGL(k_mag_pack, dim3((wp / 2 + 3) / 4, (height + 7) / 8), dim3(256), 0, ctx->stream, d_flow,
                           flow_stride_floats, width, height, thresh, a, wp);
    }
    NSOF_HIP(ctx, hipGetLastError());
    // (dilate, erode) x iterations; with no iterations the chain is a single identity pass (1x1 element)
    if (iterations == 0) {
        uint8_t one = 1;
        if ((rc = build_elem(ctx, &one, 1, 1, 0, 0, &el))) return rc;
        return run_chain(ctx, a, b, width, height, el, 1, 1u, d_mask, mask_stride);
    }
    unsigned ops = 0;
    for (int k = 0; k < iterations; k++) ops |= 1u << (2 * k);   // even passes dilate, odd passes erode
    return run_chain(ctx, a, b, width, height, el, 2 * iterations, ops, d_mask, mask_stride);
}

extern "C" int nsof_motion_mask(nsof_ctx* ctx, const float* flow, ptrdiff_t flow_stride_bytes, int width, int height,
                                double thresh, int ksize, int iterations, uint8_t* mask, ptrdiff_t mask_stride)
{
    if (!ctx) return NSOF_EINVAL;
    if (!flow || !mask) return nsof_set_error(ctx, NSOF_EINVAL, "null pointer");
    if (width < 1 || height < 1) return nsof_set_error(ctx, NSOF_ESHAPE, "empty flow field");
    if (flow_stride_bytes < (ptrdiff_t)width * 8 || mask_stride < width)
        return nsof_set_error(ctx, NSOF_EINVAL, "bad stride");
    NSOF_HIP(ctx, hipSetDevice(ctx->device));
    const size_t n0 = (size_t)width * height;
    const size_t szF = (n0 * 8 + 255) & ~(size_t)255, szM = (n0 + 255) & ~(size_t)255;
    int rc;
    if ((rc = nsof_ws_reserve(ctx, &ctx->stage, &ctx->stage_bytes, szF + szM))) return rc;
    if ((rc = nsof_hstage_reserve(ctx, szF + szM))) return rc;
    float* hF = (float*)ctx->hstage;
    uint8_t* hM = (uint8_t*)ctx->hstage + szF;
    float* dF = (float*)ctx->stage;
    uint8_t* dM = (uint8_t*)ctx->stage + szF;
    const bool in_dense = flow_stride_bytes == (ptrdiff_t)width * 8, out_dense = mask_stride == width;
    if (!in_dense)
        for (int y = 0; y < height; y++)
            memcpy(hF + (size_t)y * width * 2, (const char*)flow + (ptrdiff_t)y * flow_stride_bytes, (size_t)width * 8);
    NSOF_HIP(ctx, hipMemcpyAsync(dF, in_dense ? flow : hF, n0 * 8, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = nsof_motion_mask_dev(ctx, dF, 2 * (ptrdiff_t)width, width, height, thresh, ksize, iterations, dM, width)))
        return rc;
    NSOF_HIP(ctx, hipMemcpyAsync(out_dense ? mask : hM, dM, n0, hipMemcpyDeviceToHost, ctx->stream));
    if (int rcs = nsof_stream_sync_checked(ctx)) return rcs;   // incl. a lost hand-over of the exact-order flow kernels
    if (!out_dense)
        for (int y = 0; y < height; y++) memcpy(mask + (ptrdiff_t)y * mask_stride, hM + (size_t)y * width, (size_t)width);
    return NSOF_OK;
}
