// Exact-order Farneback iteration for SMALL batches (a lone frame pair, a handful of ROI crops): latency first.
//
// The single-kernel form (k_iterate_x) gives a (strip, image) job to one workgroup that walks the whole image height;
// a lone 1920x1080 pair is 10 such jobs on 256 CUs and one iteration takes ~0.6 ms however idle the chip is.  The
// library's summation order is sequential along y (column sums) and along x (row sums) -- but only THOSE recurrences
// are: one double addition per row / column.  So for a batch too small to fill the chip the iteration is split into
// three kernels, each as wide as its step allows, with the intermediates in HBM (they stay in the 256 MB MALL):
//   k_lat_matrices   thread <-> pixel          FarnebackUpdateMatrices                       M [5][h][w] f32
//   k_lat_colsum     thread <-> (column, plane) the library's running column sums, top down   V [h][5][w] f64
//   k_rowscan_solve  thread <-> (row, plane)    the library's running row sums + 2x2 solve    (farneback_iterate.hip)
// Same arithmetic, same order, same bits as k_iterate_x (upstream FarnebackUpdateFlow_Blur, optflowgf.cpp); 60 B/px of extra HBM traffic, which is why large batches keep the fused kernel.
#include <hip/hip_runtime.h>

#include "iterate_common.h"

namespace {

template <bool HET>
__global__ __launch_bounds__(256) void k_lat_matrices(const float* __restrict__ R0b, const float* __restrict__ R1b,
                                                       size_t pair_stride, const float* __restrict__ flow_in, int W, int H,
                                                       float* __restrict__ M, const nsof_het_item* __restrict__ items)
{
    if constexpr (HET) {
        const nsof_het_item& it = items[blockIdx.z];
        W = it.wk;
        H = it.hk;
        R0b += it.offR;
        R1b = R0b + 5 * (size_t)W * H;
        flow_in += 2 * it.offF;
        M += it.offR / 2;
    } else {
        const size_t plane = (size_t)W * H;
        R0b += (size_t)blockIdx.z * pair_stride;
        R1b += (size_t)blockIdx.z * pair_stride;
        flow_in += (size_t)blockIdx.z * plane * 2;
        M += (size_t)blockIdx.z * plane * 5;
    }
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const size_t plane = (size_t)W * H;
    const Planes R0 = planes_of(R0b, plane), R1 = planes_of(R1b, plane);
    const size_t pix = (size_t)y * W + x;
    RowIn in;
    issue_row(in, R0, R1, W, H, x, y, reinterpret_cast<const float2*>(flow_in)[pix]);
    float m5[5];
    matrix_from(in, x, y, W, H, m5);
#pragma unroll
    for (int c = 0; c < 5; c++) M[c * plane + pix] = m5[c];
}

// The library's column sums (FarnebackUpdateFlow_Blur's vsum rows): vsum = float(M[0] * (m + 2)), += M[y] for y = 1..m-1,
// then per row += double(float(M[y + m] - M[y - m - 1])) with replicated borders.  One double addition per row is the
// whole recurrence; the loads and float differences of LAT_U rows are formed ahead of it (two register sets).
constexpr int LAT_U = 16;
template <bool HET>
__global__ __launch_bounds__(64) void k_lat_colsum(const float* __restrict__ M, int W, int H, int m, double* __restrict__ V,
                                                    const nsof_het_item* __restrict__ items)
{
    if constexpr (HET) {
        const nsof_het_item& it = items[blockIdx.z];
        W = it.wk;
        H = it.hk;
        M += it.offR / 2;
        V += it.offR / 2;
    } else {
        M += (size_t)blockIdx.z * 5 * W * H;
        V += (size_t)blockIdx.z * 5 * W * H;
    }
    const int x = blockIdx.x * 64 + threadIdx.x, c = blockIdx.y;
    if (x >= W) return;
    const float* Mc = M + (size_t)c * W * H + x;
    double* Vc = V + (size_t)c * W + x;          // V[(y * 5 + c) * W + x]
    const size_t vrow = (size_t)5 * W;
    double vs = (double)(Mc[0] * (float)(m + 2));
    for (int y = 1; y < m; y++) vs += (double)Mc[(size_t)min(y, H - 1) * W];
    float d[LAT_U], nd[LAT_U];
    auto fetch = [&](int y0, float (&o)[LAT_U]) {
#pragma unroll
        for (int j = 0; j < LAT_U; j++) {
            const int y = y0 + j;
            o[j] = Mc[(size_t)min(y + m, H - 1) * W] - Mc[(size_t)min(max(y - m - 1, 0), H - 1) * W];
        }
    };
    fetch(0, d);
    for (int y0 = 0; y0 < H; y0 += LAT_U) {
        if (y0 + LAT_U < H) fetch(y0 + LAT_U, nd);
#pragma unroll
        for (int j = 0; j < LAT_U; j++) {
            if (y0 + j < H) {
                vs += (double)d[j];
                Vc[(size_t)(y0 + j) * vrow] = vs;
            }
        }
#pragma unroll
        for (int j = 0; j < LAT_U; j++) d[j] = nd[j];
    }
}

}  // namespace

// M: 5 floats per pixel, V: 5 doubles per pixel (per pair; work list: at offR / 2 of each item, as the two-kernel form).
int nsof_launch_iterate_lat(nsof_ctx* ctx, int n_pairs, const float* R0, const float* R1, size_t pair_stride,
                            const float* flow_in, float* flow_out, int W, int H, int winsize, float* M, double* V)
{
    {
        nsof_prof_scope ps(ctx, NSOF_K_ITERATE);
        hipLaunchKernelGGL(k_lat_matrices<false>, dim3((W + 63) / 64, (H + 3) / 4, n_pairs), dim3(256), 0, ctx->stream, R0, R1,
                           pair_stride, flow_in, W, H, M, nullptr);
        hipLaunchKernelGGL(k_lat_colsum<false>, dim3((W + 63) / 64, 5, n_pairs), dim3(64), 0, ctx->stream, (const float*)M, W, H,
                           winsize / 2, V, nullptr);
    }
    return nsof_launch_rowscan_solve(ctx, n_pairs, V, W, H, winsize, flow_out);
}

int nsof_launch_iterate_lat_het(nsof_ctx* ctx, int n_items, const nsof_het_item* d_items, int max_w, int max_h, const float* R,
                                const float* flow_in, float* flow_out, bool final, int winsize, float* M, double* V)
{
    {
        nsof_prof_scope ps(ctx, NSOF_K_ITERATE);
        hipLaunchKernelGGL(k_lat_matrices<true>, dim3((max_w + 63) / 64, (max_h + 3) / 4, n_items), dim3(256), 0, ctx->stream, R, R,
                           (size_t)0, flow_in, 0, 0, M, d_items);
        hipLaunchKernelGGL(k_lat_colsum<true>, dim3((max_w + 63) / 64, 5, n_items), dim3(64), 0, ctx->stream, (const float*)M, 0, 0,
                           winsize / 2, V, d_items);
    }
    return nsof_launch_rowscan_solve_het(ctx, n_items, d_items, max_h, V, flow_out, final, winsize);
}
