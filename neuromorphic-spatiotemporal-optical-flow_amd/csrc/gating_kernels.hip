// ROI gating on the device: the reference thresholds a tiny map of device currents, labels its
// 4-connected components and turns their bounding boxes into crop rectangles
// (/root/reference/optical_flow_seg.py:115-121 update_transition_pic, :211-252 opticalFlow3D, :426-431 current -> gray).
// Here ONE wavefront owns one gating map (up to 64 x 64 cells; the reference's are 4 x 4 .. 13 x 24):
//   lane r      holds row r of the thresholded map as a 64-bit mask (bit c = column c);
//   components  are taken in raster order of their first cell (the label order of cv2.connectedComponentsWithStats and
//               of the host mirror nsof_roi_from_surface): seed = first set bit of the first non-empty row, then a flood
//               fill by mask arithmetic -- S |= (S << 1 | S >> 1 | S of the row above | S of the row below) & R, the
//               neighbouring rows through wave shuffles -- until no lane changes (a ballot);
//   boxes       top / bottom from a ballot of the non-empty rows, left / right from the OR of all rows (xor-shuffle
//               reduction); scaled by MEMSIZE, extended and clipped exactly as the reference does.
// Output per map: the number of rectangles and rects[cap][4] = (x0, y0, x1, y1), FLAG 1 one per component, FLAG 2 their union.
#include <cmath>

#include "nsof_internal.h"

namespace {

__device__ __forceinline__ unsigned long long wave_or(unsigned long long v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v |= __shfl_xor(v, o);
    return v;
}

__global__ __launch_bounds__(64) void k_roi_gate(const double* __restrict__ cur, size_t map_stride, int rows, int cols, int fw,
                                                 int fh, int ms, int thres, int el, int er, int eu, int ed, int conn8, int flag,
                                                 int cap, int* __restrict__ counts, int* __restrict__ rects,
                                                 unsigned char* __restrict__ gray)
{
    const int k = blockIdx.x, lane = threadIdx.x;
    const double* m = cur + (size_t)k * map_stride;
    unsigned long long R = 0;
    if (lane < rows) {
        for (int c = 0; c < cols; c++) {
            double g = -3366.0 / log10(m[(size_t)lane * cols + c]) - 306.0;
            g = g < 0.0 ? 0.0 : (g > 255.0 ? 255.0 : g);        // NaN (I <= 0) compares false twice and casts to 0
            const int gi = (g == g) ? (int)(unsigned char)g : 0;
            if (gray) gray[((size_t)k * rows + lane) * cols + c] = (unsigned char)gi;
            if (gi >= thres) R |= 1ull << c;
        }
    }
    int* out = rects + (size_t)k * cap * 4;
    int n = 0, ux0 = 1 << 30, uy0 = 1 << 30, ux1 = -1, uy1 = -1;
    auto emit = [&](int bx0, int by0, int bx1, int by1) {   // cell box (inclusive) -> frame rectangle
        if (lane == 0 && n < cap) {
            out[4 * n] = max(bx0 * ms - el, 0);
            out[4 * n + 1] = max(by0 * ms - eu, 0);
            out[4 * n + 2] = min((bx1 + 1) * ms + er, fw);
            out[4 * n + 3] = min((by1 + 1) * ms + ed, fh);
        }
        n++;
    };
    for (int guard = 0; guard < 64 * 64; guard++) {   // at most one component per cell
        const unsigned long long any = __ballot(R != 0);
        if (!any) break;
        const int r0 = __ffsll((long long)any) - 1;
        const unsigned long long Rr0 = __shfl(R, r0);
        const int c0 = __ffsll((long long)Rr0) - 1;
        unsigned long long S = lane == r0 ? (1ull << c0) : 0;
        for (int it = 0; it < 64 * 64; it++) {
            unsigned long long up = __shfl_up(S, 1), dn = __shfl_down(S, 1);
            if (lane == 0) up = 0;
            if (lane == 63) dn = 0;
            unsigned long long N = S | (S << 1) | (S >> 1) | up | dn;
            if (conn8) N |= (up << 1) | (up >> 1) | (dn << 1) | (dn >> 1);
            N &= R;
            if (!__ballot(N != S)) break;
            S = N;
        }
        const unsigned long long rmask = __ballot(S != 0), cmask = wave_or(S);
        const int by0 = __ffsll((long long)rmask) - 1, by1 = 63 - __clzll((long long)rmask);
        const int bx0 = __ffsll((long long)cmask) - 1, bx1 = 63 - __clzll((long long)cmask);
        if (flag == 1) {
            emit(bx0, by0, bx1, by1);
        } else {
            ux0 = min(ux0, bx0); uy0 = min(uy0, by0); ux1 = max(ux1, bx1); uy1 = max(uy1, by1);
        }
        R &= ~S;
    }
    if (flag == 2 && ux1 >= 0) emit(ux0, uy0, ux1, uy1);
    if (lane == 0) counts[k] = n;
}

}  // namespace

// Device twin of nsof_roi_from_surface for n_maps maps at once; everything stays on the context's stream, nothing is
// synchronised.  d_gray (optional): the 8-bit gating maps, [n_maps][rows][cols].
extern "C" int nsof_roi_from_surface_dev(nsof_ctx* ctx, const double* d_current, int n_maps, size_t map_stride, int rows, int cols,
                                         int frame_w, int frame_h, int memsize, int thres, int extend_left, int extend_right,
                                         int extend_upper, int extend_lower, int connectivity, int flag, int max_rects,
                                         int* d_counts, int* d_rects, unsigned char* d_gray)
{
    if (!ctx) return NSOF_EINVAL;
    if (!d_current || !d_counts || !d_rects || n_maps < 1 || rows < 1 || cols < 1 || frame_w < 1 || frame_h < 1 || memsize < 1 ||
        (connectivity != 4 && connectivity != 8) || (flag != 1 && flag != 2) || max_rects < 1 || map_stride < (size_t)rows * cols)
        return nsof_set_error(ctx, NSOF_EINVAL, "bad gating arguments");
    if (rows > 64 || cols > 64) return nsof_set_error(ctx, NSOF_EUNSUPPORTED, "gating maps of at most 64 x 64 cells (got %d x %d)", rows, cols);
    if (rows > frame_h / memsize || cols > frame_w / memsize)   // the reference's loop would write outside its transition picture
        return nsof_set_error(ctx, NSOF_ESHAPE, "gating map %d x %d larger than the frame's %d x %d blocks", rows, cols,
                              frame_h / memsize, frame_w / memsize);
    NSOF_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_roi_gate, dim3(n_maps), dim3(64), 0, ctx->stream, d_current, map_stride, rows, cols, frame_w, frame_h,
                       memsize, thres, extend_left, extend_right, extend_upper, extend_lower, connectivity == 8 ? 1 : 0, flag,
                       max_rects, d_counts, d_rects, d_gray);
    NSOF_HIP(ctx, hipGetLastError());
    return NSOF_OK;
}
