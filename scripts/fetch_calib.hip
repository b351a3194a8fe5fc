// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access widths this repo's kernels use (the guide calibrates the
// counter for 16-B-per-lane streaming reads only: "other access widths are uncalibrated: calibrate on a known byte count in
// your own access pattern").  Each kernel streams a 1 GiB buffer (4x the 256 MiB Infinity Cache) ONCE with one load of
// 4 / 8 / 16 bytes per lane, lanes consecutive -- the patterns of k_prep_same3_vec / k_polyexp (dword per lane), the flow
// reads (8 B) and the R gathers (16 B).  Known bytes / (FETCH_SIZE x 1024) = the multiplier scripts/prof_traffic*.sh apply.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/fetch_calib scripts/fetch_calib.hip
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/fetch_calib -- scripts/fetch_calib
#include <hip/hip_runtime.h>

#include <cstdio>

template <typename T>
__global__ __launch_bounds__(256) void k_stream_read(const T* __restrict__ p, size_t n, unsigned* sink)
{
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        T v = p[i];
        const unsigned* w = reinterpret_cast<const unsigned*>(&v);
#pragma unroll
        for (unsigned k = 0; k < sizeof(T) / 4; k++) acc ^= w[k];
    }
    if (acc == 0x12345678u) *sink = acc;   // never true for the fill pattern: keeps the loads alive
}

// The level-0 pyramid kernel's pattern (k_prep_same3_vec): 8-bit frames of 1920 x 1080, a wave reads 10 source rows (its 8 + one
// above and below) of a 256-byte row segment, one dword per lane: it REQUESTS 1.25x the frame bytes; what reaches the fabric
// depends on how much of the overlap the L2 absorbs.
__global__ __launch_bounds__(256) void k_rows_10_per_8(const unsigned char* __restrict__ img, int W, int H, unsigned* sink)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = (blockIdx.x * 64 + lane) * 4;
    const int y0 = (blockIdx.y * 4 + wave) * 8;
    const unsigned char* p = img + (size_t)blockIdx.z * W * H;
    unsigned acc = 0;
    if (x < W)
        for (int r = -1; r <= 8; r++) {
            const int y = min(max(y0 + r, 0), H - 1);
            if (y0 < H) acc ^= *reinterpret_cast<const unsigned*>(p + (size_t)y * W + x);
        }
    if (acc == 0x12345678u) *sink = acc;
}

// The pattern of the vertical pass of k_polyexp_rs<.., U8>: a thread owns a column of a 240-column strip (+ 8 halo columns per
// side, so neighbouring strips overlap by 16) and walks down a 64-row segment (+ 6 rows of warm-up per side) reading the bytes
// at x - 1, x, x + 1 of every row with three byte loads.
__global__ __launch_bounds__(256) void k_bytes_3_per_px(const unsigned char* __restrict__ img, int W, int H, unsigned* sink)
{
    const int x = min(max((int)blockIdx.x * 240 - 8 + (int)threadIdx.x, 0), W - 1);
    const int xl = max(x - 1, 0), xr = min(x + 1, W - 1);
    const int y0 = blockIdx.y * 64;
    const unsigned char* p = img + (size_t)blockIdx.z * W * H;
    unsigned acc = 0;
    for (int r = y0 - 6; r < y0 + 64 + 6; r++) {
        const unsigned char* rp = p + (size_t)min(max(r, 0), H - 1) * W;
        acc += rp[xl] + rp[x] + rp[xr];
    }
    if (acc == 0x12345678u) *sink = acc;
}

int main()
{
    const size_t bytes = (size_t)1 << 30;
    void* buf = nullptr;
    unsigned* sink = nullptr;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc((void**)&sink, 4) != hipSuccess) return 1;
    hipMemset(buf, 0x5a, bytes);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k_stream_read<unsigned>, dim3(256 * 16), dim3(256), 0, 0, (const unsigned*)buf, bytes / 4, sink);
        hipLaunchKernelGGL(k_stream_read<uint2>, dim3(256 * 16), dim3(256), 0, 0, (const uint2*)buf, bytes / 8, sink);
        hipLaunchKernelGGL(k_stream_read<uint4>, dim3(256 * 16), dim3(256), 0, 0, (const uint4*)buf, bytes / 16, sink);
    }
    {
        const int W = 1920, H = 1080, n = (int)(bytes / ((size_t)W * H));   // 517 frames
        for (int rep = 0; rep < 3; rep++)
            hipLaunchKernelGGL(k_rows_10_per_8, dim3((W / 4 + 63) / 64, (H + 31) / 32, n), dim3(256), 0, 0, (const unsigned char*)buf, W, H, sink);
        for (int rep = 0; rep < 3; rep++)
            hipLaunchKernelGGL(k_bytes_3_per_px, dim3((W + 239) / 240, (H + 63) / 64, n), dim3(256), 0, 0, (const unsigned char*)buf, W, H, sink);
        printf("rows pattern: %d frames, %zu frame bytes per launch\n", n, (size_t)n * W * H);
    }
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    printf("streamed %zu bytes per launch\n", bytes);
    return 0;
}
