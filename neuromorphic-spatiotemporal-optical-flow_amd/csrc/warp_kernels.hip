// Frame-prediction task on the flow field (reference optical_flow_prediction.py:255-353, 581-591,
// 113-115):  map = grid + (-flow);  cv2.remap(next_frame channel, map, INTER_LINEAR[, BORDER_REPLICATE]);
// structural_similarity(true[:,:,2], prediction[:,:,2], data_range=255).
//
//   k_remap_u8<CN, FUSED>  8-bit bilinear remap in cv2's fixed point (1/32-pixel map, 15-bit weights).  FUSED forms
//                          the map from the flow canvas in the kernel (float32(double(x) - double(f)), as NumPy
//                          does) so the float maps never exist in memory: 8 B/px flow in, CN B/px out, plus the
//                          gathered source (read once through L2 for a smooth field).
//   k_ssim_partial/final   SSIM of one channel: exact integer 7x7 window sums (separable, through LDS), S in double,
//                          fixed-order reduction (per-block partials, then one block) -> deterministic.
#include <cmath>
#include <cstring>

#include "nsof_internal.h"

namespace {

struct MapSrc {
    const float* a;      // FUSED: flow canvas (u,v interleaved) ; else map_x
    const float* b;      // else map_y
    ptrdiff_t astride;   // row stride in floats
    ptrdiff_t bstride;
    int x0, y0;          // FUSED: position of the destination crop inside the canvas
    int ox, oy;          // FUSED: canvas position of the first vector stored at `a` (0,0 for a whole canvas)
    int sign;            // FUSED: map = grid + sign * flow
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// One thread per destination pixel, all CN interleaved channels.
template <int CN, bool FUSED>
__global__ __launch_bounds__(256) void k_remap_u8(const uint8_t* __restrict__ src, ptrdiff_t sstride, int sw, int sh,
                                                  MapSrc m, int dw, int dh, int border, int cval,
                                                  uint8_t* __restrict__ dst, ptrdiff_t dstride)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= dw || y >= dh) return;
    float mx, my;
    if (FUSED) {
        const float2 f = *(const float2*)(m.a + (ptrdiff_t)(m.y0 + y - m.oy) * m.astride + 2 * (m.x0 + x - m.ox));
        mx = (float)((double)(m.x0 + x) + (double)m.sign * (double)f.x);
        my = (float)((double)(m.y0 + y) + (double)m.sign * (double)f.y);
    } else {
        mx = m.a[(ptrdiff_t)y * m.astride + x];
        my = m.b[(ptrdiff_t)y * m.bstride + x];
    }
    // cvRound(v * 32): the product is exact, rintf rounds half to even; saturate like the float -> int conversion
    const int lx = (int)fminf(fmaxf(rintf(mx * 32.0f), -2147483648.0f), 2147483520.0f);
    const int ly = (int)fminf(fmaxf(rintf(my * 32.0f), -2147483648.0f), 2147483520.0f);
    const int fx = lx & 31, fy = ly & 31;
    const int ix = clampi(lx >> 5, -32768, 32767), iy = clampi(ly >> 5, -32768, 32767);
    const int w0 = (32 - fx) * (32 - fy) * 32, w1 = fx * (32 - fy) * 32, w2 = (32 - fx) * fy * 32, w3 = fx * fy * 32;
    uint8_t* o = dst + (ptrdiff_t)y * dstride + x * CN;
    int x0, x1, y0, y1;
    bool in00 = true, in01 = true, in10 = true, in11 = true;
    if (border == 1) {   // BORDER_REPLICATE: clamp the tap coordinates
        x0 = clampi(ix, 0, sw - 1); x1 = clampi(ix + 1, 0, sw - 1);
        y0 = clampi(iy, 0, sh - 1); y1 = clampi(iy + 1, 0, sh - 1);
    } else {             // BORDER_CONSTANT
        if (ix >= sw || ix + 1 < 0 || iy >= sh || iy + 1 < 0) {
#pragma unroll
            for (int c = 0; c < CN; c++) o[c] = (uint8_t)cval;
            return;
        }
        const bool xi0 = ix >= 0, xi1 = ix + 1 < sw, yi0 = iy >= 0, yi1 = iy + 1 < sh;
        in00 = xi0 && yi0; in01 = xi1 && yi0; in10 = xi0 && yi1; in11 = xi1 && yi1;
        x0 = clampi(ix, 0, sw - 1); x1 = clampi(ix + 1, 0, sw - 1);
        y0 = clampi(iy, 0, sh - 1); y1 = clampi(iy + 1, 0, sh - 1);
    }
    const uint8_t* r0 = src + (ptrdiff_t)y0 * sstride;
    const uint8_t* r1 = src + (ptrdiff_t)y1 * sstride;
    if (CN == 3 && in00 && in01 && in10 && in11 && x1 == x0 + 1 && x0 + 3 <= sw) {
        // interior: the two taps of a row are 6 adjacent bytes -> one unaligned 8-byte load per row instead of 6
        // byte loads (the L1 serves 4 lanes per cycle per instruction, so the instruction count is what costs)
        unsigned long long t, b;
        __builtin_memcpy(&t, r0 + x0 * 3, 8);
        __builtin_memcpy(&b, r1 + x0 * 3, 8);
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const int v0 = (int)((t >> (8 * c)) & 0xffu), v1 = (int)((t >> (8 * (c + 3))) & 0xffu);
            const int v2 = (int)((b >> (8 * c)) & 0xffu), v3 = (int)((b >> (8 * (c + 3))) & 0xffu);
            const int r = (v0 * w0 + v1 * w1 + v2 * w2 + v3 * w3 + (1 << 14)) >> 15;
            o[c] = (uint8_t)clampi(r, 0, 255);
        }
        return;
    }
#pragma unroll
    for (int c = 0; c < CN; c++) {
        const int v0 = in00 ? r0[x0 * CN + c] : cval, v1 = in01 ? r0[x1 * CN + c] : cval;
        const int v2 = in10 ? r1[x0 * CN + c] : cval, v3 = in11 ? r1[x1 * CN + c] : cval;
        const int r = (v0 * w0 + v1 * w1 + v2 * w2 + v3 * w3 + (1 << 14)) >> 15;
        o[c] = (uint8_t)clampi(r, 0, 255);
    }
}

// ---- SSIM ---------------------------------------------------------------------------------------------------
constexpr int SX = 64, SY = 16, WIN = 7, PAD = 3;

__global__ __launch_bounds__(256) void k_ssim_partial(const uint8_t* __restrict__ a, ptrdiff_t astride, int aps,
                                                      const uint8_t* __restrict__ b, ptrdiff_t bstride, int bps, int w,
                                                      int h, double c1, double c2, double* __restrict__ partial)
{
    __shared__ uint8_t ta[SY + 6][SX + 6], tb[SY + 6][SX + 6];
    __shared__ int hs[5][SY + 6][SX];     // horizontal 7-sums of a, b, a*a, b*b, a*b
    __shared__ double wsum[4];
    const int tid = threadIdx.x;
    const int ox = blockIdx.x * SX + PAD, oy = blockIdx.y * SY + PAD;   // first output pixel of the tile
    for (int i = tid; i < (SY + 6) * (SX + 6); i += 256) {
        const int r = i / (SX + 6), c = i - r * (SX + 6);
        const int gy = oy - PAD + r, gx = ox - PAD + c;
        const bool in = gy < h && gx < w;
        ta[r][c] = in ? a[(ptrdiff_t)gy * astride + (ptrdiff_t)gx * aps] : 0;
        tb[r][c] = in ? b[(ptrdiff_t)gy * bstride + (ptrdiff_t)gx * bps] : 0;
    }
    __syncthreads();
    for (int i = tid; i < (SY + 6) * SX; i += 256) {
        const int r = i / SX, c = i - r * SX;
        int s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0;
#pragma unroll
        for (int k = 0; k < WIN; k++) {
            const int p = ta[r][c + k], q = tb[r][c + k];
            s0 += p; s1 += q; s2 += p * p; s3 += q * q; s4 += p * q;
        }
        hs[0][r][c] = s0; hs[1][r][c] = s1; hs[2][r][c] = s2; hs[3][r][c] = s3; hs[4][r][c] = s4;
    }
    __syncthreads();
    const double NP = WIN * WIN, cov_norm = NP / (NP - 1);
    double acc = 0;
    for (int i = tid; i < SY * SX; i += 256) {
        const int r = i / SX, c = i - r * SX;
        if (oy + r >= h - PAD || ox + c >= w - PAD) continue;
        int s[5] = {0, 0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < WIN; k++)
#pragma unroll
            for (int q = 0; q < 5; q++) s[q] += hs[q][r + k][c];
        const double ux = s[0] / NP, uy = s[1] / NP, uxx = s[2] / NP, uyy = s[3] / NP, uxy = s[4] / NP;
        const double vx = cov_norm * (uxx - ux * ux), vy = cov_norm * (uyy - uy * uy), vxy = cov_norm * (uxy - ux * uy);
        const double A1 = 2 * ux * uy + c1, A2 = 2 * vxy + c2, B1 = ux * ux + uy * uy + c1, B2 = vx + vy + c2;
        acc += (A1 * A2) / (B1 * B2);
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((tid & 63) == 0) wsum[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) partial[blockIdx.y * gridDim.x + blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

__global__ __launch_bounds__(256) void k_ssim_final(const double* __restrict__ partial, int n, double count,
                                                    double* __restrict__ out)
{
    __shared__ double red[256];
    double acc = 0;
    for (int i = threadIdx.x; i < n; i += 256) acc += partial[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = red[0] / count;
}

template <bool FUSED>
int launch_remap(nsof_ctx* ctx, int cn, const uint8_t* src, ptrdiff_t sstride, int sw, int sh, const MapSrc& m, int dw,
                 int dh, int border, int cval, uint8_t* dst, ptrdiff_t dstride)
{
    dim3 grid((dw + 63) / 64, (dh + 3) / 4);
    nsof_prof_scope ps(ctx, NSOF_K_REMAP);
    if (cn == 1)
        hipLaunchKernelGGL((k_remap_u8<1, FUSED>), grid, dim3(256), 0, ctx->stream, src, sstride, sw, sh, m, dw, dh,
                           border, cval, dst, dstride);
    else
        hipLaunchKernelGGL((k_remap_u8<3, FUSED>), grid, dim3(256), 0, ctx->stream, src, sstride, sw, sh, m, dw, dh,
                           border, cval, dst, dstride);
    NSOF_HIP(ctx, hipGetLastError());
    return NSOF_OK;
}

int check_remap_args(nsof_ctx* ctx, int cn, int sw, int sh, ptrdiff_t sstride, int dw, int dh, ptrdiff_t dstride,
                     int border)
{
    if (cn != 1 && cn != 3) return nsof_set_error(ctx, NSOF_EUNSUPPORTED, "remap: 1 or 3 interleaved channels");
    if (sw < 1 || sh < 1 || dw < 1 || dh < 1) return nsof_set_error(ctx, NSOF_ESHAPE, "remap: empty image");
    if (sw > 32767 || sh > 32767) return nsof_set_error(ctx, NSOF_EUNSUPPORTED, "remap: source above 32767 px");
    if (sstride < (ptrdiff_t)sw * cn || dstride < (ptrdiff_t)dw * cn) return nsof_set_error(ctx, NSOF_EINVAL, "stride");
    if (border != NSOF_BORDER_CONSTANT && border != NSOF_BORDER_REPLICATE)
        return nsof_set_error(ctx, NSOF_EUNSUPPORTED, "remap: borderMode must be BORDER_CONSTANT or BORDER_REPLICATE");
    return NSOF_OK;
}

}  // namespace

// cv2.cvtColor(frame, COLOR_RGB2GRAY / COLOR_BGR2GRAY) for 8-bit frames: (c0*w0 + c1*w1 + c2*w2 + 2^14) >> 15 with the
// fixed-point weights 9798 / 19235 / 3735 (optical_flow_seg.py:442-447 applies RGB2GRAY to imread's B,G,R frames, so
// blue takes the red weight; both orders are provided).  A lane converts 4 pixels: 12 source bytes -> one dword.
__global__ __launch_bounds__(256) void k_gray_u8(const uint8_t* __restrict__ src, ptrdiff_t src_stride, int W, int H, int w0,
                                                  int w1, int w2, uint8_t* __restrict__ dst, ptrdiff_t dst_stride)
{
    const int x = (blockIdx.x * 256 + threadIdx.x) * 4, y = blockIdx.y;
    if (x >= W || y >= H) return;
    const uint8_t* s = src + (ptrdiff_t)y * src_stride + 3 * (ptrdiff_t)x;
    uint8_t* d = dst + (ptrdiff_t)y * dst_stride + x;
    const int n = min(4, W - x);
    unsigned packed = 0;
    for (int i = 0; i < n; i++) {
        const unsigned g = ((unsigned)s[3 * i] * w0 + (unsigned)s[3 * i + 1] * w1 + (unsigned)s[3 * i + 2] * w2 + (1u << 14)) >> 15;
        packed |= g << (8 * i);
    }
    if (n == 4 && (reinterpret_cast<uintptr_t>(d) & 3) == 0) {
        *reinterpret_cast<unsigned*>(d) = packed;
    } else {
        for (int i = 0; i < n; i++) d[i] = (uint8_t)(packed >> (8 * i));
    }
}

extern "C" int nsof_gray_u8_dev(nsof_ctx* ctx, const uint8_t* d_src, ptrdiff_t src_stride, int width, int height, int bgr_weights,
                                uint8_t* d_dst, ptrdiff_t dst_stride)
{
    if (!ctx || !d_src || !d_dst || width < 1 || height < 1 || src_stride < 3 * (ptrdiff_t)width || dst_stride < width)
        return NSOF_EINVAL;
    NSOF_HIP(ctx, hipSetDevice(ctx->device));
    nsof_prof_scope ps(ctx, NSOF_K_REMAP);
    const int w0 = bgr_weights ? 3735 : 9798, w2 = bgr_weights ? 9798 : 3735;
    hipLaunchKernelGGL(k_gray_u8, dim3((width + 1023) / 1024, height), dim3(256), 0, ctx->stream, d_src, src_stride, width, height,
                       w0, 19235, w2, d_dst, dst_stride);
    NSOF_HIP(ctx, hipGetLastError());
    return NSOF_OK;
}

extern "C" int nsof_remap_linear_u8_dev(nsof_ctx* ctx, const uint8_t* d_src, ptrdiff_t src_stride, int src_w, int src_h,
                                        int channels, const float* d_map_x, ptrdiff_t map_x_stride_floats,
                                        const float* d_map_y, ptrdiff_t map_y_stride_floats, int dst_w, int dst_h,
                                        int border_mode, int border_value, uint8_t* d_dst, ptrdiff_t dst_stride)
{
    if (!ctx) return NSOF_EINVAL;
    if (!d_src || !d_map_x || !d_map_y || !d_dst) return nsof_set_error(ctx, NSOF_EINVAL, "null pointer");
    int rc = check_remap_args(ctx, channels, src_w, src_h, src_stride, dst_w, dst_h, dst_stride, border_mode);
    if (rc) return rc;
    if (map_x_stride_floats < dst_w || map_y_stride_floats < dst_w) return nsof_set_error(ctx, NSOF_EINVAL, "map stride");
    NSOF_HIP(ctx, hipSetDevice(ctx->device));
    MapSrc m{d_map_x, d_map_y, map_x_stride_floats, map_y_stride_floats, 0, 0, 0, 0, 0};
    return launch_remap<false>(ctx, channels, d_src, src_stride, src_w, src_h, m, dst_w, dst_h, border_mode,
                               border_value & 255, d_dst, dst_stride);
}

extern "C" int nsof_predict_warp_u8_dev(nsof_ctx* ctx, const uint8_t* d_frame, ptrdiff_t frame_stride, int width,
                                        int height, int channels, const float* d_flow, ptrdiff_t flow_stride_floats,
                                        int sign, int x0, int y0, int x1, int y1, int border_mode, uint8_t* d_out,
                                        ptrdiff_t out_stride)
{
    if (!ctx) return NSOF_EINVAL;
    if (!d_frame || !d_flow || !d_out) return nsof_set_error(ctx, NSOF_EINVAL, "null pointer");
    if (x0 < 0 || y0 < 0 || x1 > width || y1 > height || x1 <= x0 || y1 <= y0)
        return nsof_set_error(ctx, NSOF_ESHAPE, "predict_warp: region (%d,%d)-(%d,%d) outside %dx%d", x0, y0, x1, y1,
                              width, height);
    int rc = check_remap_args(ctx, channels, width, height, frame_stride, x1 - x0, y1 - y0, out_stride, border_mode);
    if (rc) return rc;
    if (sign != 1 && sign != -1) return nsof_set_error(ctx, NSOF_EINVAL, "sign must be +1 or -1");
    if (flow_stride_floats < 2 * (ptrdiff_t)width || (flow_stride_floats & 1))
        return nsof_set_error(ctx, NSOF_EINVAL, "flow stride");
    NSOF_HIP(ctx, hipSetDevice(ctx->device));
    MapSrc m{d_flow, nullptr, flow_stride_floats, 0, x0, y0, 0, 0, sign};
    // the destination is the same crop of the output frame
    return launch_remap<true>(ctx, channels, d_frame, frame_stride, width, height, m, x1 - x0, y1 - y0, border_mode, 0,
                              d_out + (ptrdiff_t)y0 * out_stride + (ptrdiff_t)x0 * channels, out_stride);
}

extern "C" int nsof_predict_warp_u8(nsof_ctx* ctx, const uint8_t* frame, ptrdiff_t frame_stride, int width, int height,
                                    int channels, const float* flow_crop, ptrdiff_t flow_stride_bytes, int sign, int x0,
                                    int y0, int x1, int y1, int border_mode, uint8_t* out, ptrdiff_t out_stride)
{
    if (!ctx) return NSOF_EINVAL;
    if (!frame || !flow_crop || !out) return nsof_set_error(ctx, NSOF_EINVAL, "null pointer");
    if (x0 < 0 || y0 < 0 || x1 > width || y1 > height || x1 <= x0 || y1 <= y0)
        return nsof_set_error(ctx, NSOF_ESHAPE, "predict_warp: region (%d,%d)-(%d,%d) outside %dx%d", x0, y0, x1, y1,
                              width, height);
    if (channels != 1 && channels != 3) return nsof_set_error(ctx, NSOF_EUNSUPPORTED, "1 or 3 channels");
    const int rw = x1 - x0, rh = y1 - y0;
    if (frame_stride < (ptrdiff_t)width * channels || out_stride < (ptrdiff_t)width * channels ||
        flow_stride_bytes < (ptrdiff_t)rw * 8)
        return nsof_set_error(ctx, NSOF_EINVAL, "stride");
    int rc;
    if ((rc = check_remap_args(ctx, channels, width, height, (ptrdiff_t)width * channels, rw, rh,
                               (ptrdiff_t)rw * channels, border_mode)))
        return rc;
    if (sign != 1 && sign != -1) return nsof_set_error(ctx, NSOF_EINVAL, "sign must be +1 or -1");
    NSOF_HIP(ctx, hipSetDevice(ctx->device));
    const size_t rowb = (size_t)width * channels;
    const size_t szI = (rowb * height + 255) & ~(size_t)255, szF = ((size_t)rw * rh * 8 + 255) & ~(size_t)255;
    const size_t szO = ((size_t)rw * rh * channels + 255) & ~(size_t)255;
    if ((rc = nsof_ws_reserve(ctx, &ctx->stage, &ctx->stage_bytes, szI + szF + szO))) return rc;
    if ((rc = nsof_hstage_reserve(ctx, szI + szF + szO))) return rc;
    uint8_t* hI = (uint8_t*)ctx->hstage;
    float* hF = (float*)(hI + szI);
    uint8_t* hO = hI + szI + szF;
    uint8_t* dI = (uint8_t*)ctx->stage;
    float* dF = (float*)(dI + szI);
    uint8_t* dO = dI + szI + szF;
    const bool dense = frame_stride == (ptrdiff_t)rowb;
    if (!dense)
        for (int y = 0; y < height; y++) memcpy(hI + (size_t)y * rowb, frame + (ptrdiff_t)y * frame_stride, rowb);
    NSOF_HIP(ctx, hipMemcpyAsync(dI, dense ? frame : hI, rowb * height, hipMemcpyHostToDevice, ctx->stream));
    for (int y = 0; y < rh; y++)
        memcpy(hF + (size_t)y * rw * 2, (const char*)flow_crop + (ptrdiff_t)y * flow_stride_bytes, (size_t)rw * 8);
    NSOF_HIP(ctx, hipMemcpyAsync(dF, hF, (size_t)rw * rh * 8, hipMemcpyHostToDevice, ctx->stream));
    // device-side flow is the crop alone: its first vector is canvas position (x0, y0)
    MapSrc m{dF, nullptr, 2 * (ptrdiff_t)rw, 0, x0, y0, x0, y0, sign};
    if ((rc = launch_remap<true>(ctx, channels, dI, (ptrdiff_t)rowb, width, height, m, rw, rh, border_mode, 0, dO,
                                 (ptrdiff_t)rw * channels)))
        return rc;
    NSOF_HIP(ctx, hipMemcpyAsync(hO, dO, (size_t)rw * rh * channels, hipMemcpyDeviceToHost, ctx->stream));
    if (int rcs = nsof_stream_sync_checked(ctx)) return rcs;   // incl. a lost hand-over of the exact-order flow kernels
    for (int y = 0; y < rh; y++)
        memcpy(out + (ptrdiff_t)(y0 + y) * out_stride + (ptrdiff_t)x0 * channels, hO + (size_t)y * rw * channels,
               (size_t)rw * channels);
    return NSOF_OK;
}

extern "C" int nsof_ssim_u8_dev(nsof_ctx* ctx, const uint8_t* d_a, ptrdiff_t a_stride, int a_pixel_step,
                                const uint8_t* d_b, ptrdiff_t b_stride, int b_pixel_step, int width, int height,
                                double data_range, double* ssim_out)
{
    if (!ctx) return NSOF_EINVAL;
    if (!d_a || !d_b || !ssim_out) return nsof_set_error(ctx, NSOF_EINVAL, "null pointer");
    if (width < WIN || height < WIN)
        return nsof_set_error(ctx, NSOF_ESHAPE, "ssim: win_size 7 exceeds the image extent %dx%d", width, height);
    if (a_pixel_step < 1 || b_pixel_step < 1 || a_stride < (ptrdiff_t)width * a_pixel_step ||
        b_stride < (ptrdiff_t)width * b_pixel_step)
        return nsof_set_error(ctx, NSOF_EINVAL, "ssim: stride");
    NSOF_HIP(ctx, hipSetDevice(ctx->device));
    const int ow = width - 2 * PAD, oh = height - 2 * PAD;
    dim3 grid((ow + SX - 1) / SX, (oh + SY - 1) / SY);
    const int nblk = grid.x * grid.y;
    int rc = nsof_ws_reserve(ctx, &ctx->tmp, &ctx->tmp_bytes, ((size_t)nblk + 1) * sizeof(double));
    if (rc) return rc;
    double* partial = (double*)ctx->tmp;
    const double c1 = (0.01 * data_range) * (0.01 * data_range), c2 = (0.03 * data_range) * (0.03 * data_range);
    {
        nsof_prof_scope ps(ctx, NSOF_K_SSIM);
        hipLaunchKernelGGL(k_ssim_partial, grid, dim3(256), 0, ctx->stream, d_a, a_stride, a_pixel_step, d_b, b_stride,
                           b_pixel_step, width, height, c1, c2, partial);
    }
    hipLaunchKernelGGL(k_ssim_final, dim3(1), dim3(256), 0, ctx->stream, partial, nblk, (double)ow * (double)oh,
                       partial + nblk);
    NSOF_HIP(ctx, hipGetLastError());
    NSOF_HIP(ctx, hipMemcpyAsync(ssim_out, partial + nblk, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (int rcs = nsof_stream_sync_checked(ctx)) return rcs;   // incl. a lost hand-over of the exact-order flow kernels
    return NSOF_OK;
}

extern "C" int nsof_ssim_u8(nsof_ctx* ctx, const uint8_t* a, ptrdiff_t a_stride, int a_pixel_step, const uint8_t* b,
                            ptrdiff_t b_stride, int b_pixel_step, int width, int height, double data_range,
                            double* ssim_out)
{
    if (!ctx) return NSOF_EINVAL;
    if (!a || !b || !ssim_out) return nsof_set_error(ctx, NSOF_EINVAL, "null pointer");
    if (width < 1 || height < 1 || a_pixel_step < 1 || b_pixel_step < 1)
        return nsof_set_error(ctx, NSOF_ESHAPE, "ssim: empty image");
    NSOF_HIP(ctx, hipSetDevice(ctx->device));
    // pack the one channel of each image into dense planes on the host side of the copy
    const size_t n0 = (size_t)width * height, sz = (n0 + 255) & ~(size_t)255;
    int rc;
    if ((rc = nsof_ws_reserve(ctx, &ctx->stage, &ctx->stage_bytes, 2 * sz))) return rc;
    if ((rc = nsof_hstage_reserve(ctx, 2 * sz))) return rc;
    uint8_t* hA = (uint8_t*)ctx->hstage;
    uint8_t* hB = hA + sz;
    for (int y = 0; y < height; y++) {
        const uint8_t* ra = a + (ptrdiff_t)y * a_stride;
        const uint8_t* rb = b + (ptrdiff_t)y * b_stride;
        for (int x = 0; x < width; x++) {
            hA[(size_t)y * width + x] = ra[(ptrdiff_t)x * a_pixel_step];
            hB[(size_t)y * width + x] = rb[(ptrdiff_t)x * b_pixel_step];
        }
    }
    NSOF_HIP(ctx, hipMemcpyAsync(ctx->stage, hA, 2 * sz, hipMemcpyHostToDevice, ctx->stream));
    return nsof_ssim_u8_dev(ctx, (const uint8_t*)ctx->stage, width, 1, (const uint8_t*)ctx->stage + sz, width, 1, width,
                            height, data_range, ssim_out);
}
