// Issue-rate probe for the instruction kinds of the polynomial-expansion kernel: one wave per SIMD slot runs a long
// chain-free stream of one instruction kind; reports cycles per wave-instruction.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 scripts/valu_rate_probe.hip -o /tmp/valu_probe && /tmp/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int KIND>
__global__ void probe(long long* out, float seed)
{
    float f0 = seed + threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3;
    double d0 = f0, d1 = f1, d2 = f2, d3 = f3;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 64; it++) {
        if (KIND == 0) { REP64(asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(seed));) }
        if (KIND == 1) { REP64(asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"((double)seed));) }
        if (KIND == 2) { REP64(asm volatile("v_fma_f64 %0, %4, %4, %0\n v_fma_f64 %1, %4, %4, %1\n v_fma_f64 %2, %4, %4, %2\n v_fma_f64 %3, %4, %4, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"((double)seed));) }
        if (KIND == 3) { REP64(asm volatile("v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(f0), "v"(f1), "v"(f2), "v"(f3));) }
        if (KIND == 4) { REP64(asm volatile("v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f32_f64 %2, %6\n v_cvt_f32_f64 %3, %7" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(d0), "v"(d1), "v"(d2), "v"(d3));) }
        if (KIND == 5) { REP64(asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"((double)seed));) }
        if (KIND == 6) { REP64(asm volatile("v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %2\n v_pk_mul_f32 %0, %0, %2\n v_pk_mul_f32 %1, %1, %2" : "+v"(d0), "+v"(d1) : "v"(d2));) }
        if (KIND == 8) { REP64(asm volatile("v_fma_f64 %0, %1, %1, %0\n v_fma_f64 %0, %1, %1, %0\n v_fma_f64 %0, %1, %1, %0\n v_fma_f64 %0, %1, %1, %0" : "+v"(d0) : "v"(d1));) }
        if (KIND == 9) { REP64(asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1" : "+v"(f0) : "v"(f1));) }
        if (KIND == 10) { REP64(asm volatile("v_add_f32 %0, %2, %3\n v_mul_f32 %0, %0, %3\n v_cvt_f64_f32 %1, %0\n v_add_f64 %4, %4, %1" : "+v"(f0), "+v"(d1), "+v"(f2) : "v"(f3), "v"(d0));) }
        if (KIND == 7) { REP64(asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(seed));) }
    }
    long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (f0 + f1 + f2 + f3 + (float)(d0 + d1 + d2 + d3) == 12345.678f) out[1] = 1;
}

template <int KIND>
void run(const char* name, long long* d_out, int waves_per_simd)
{
    long long h = 0;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<KIND>, dim3(1), dim3(256 * waves_per_simd), 0, 0, d_out, 1.0f);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(probe<KIND>, dim3(256), dim3(256 * waves_per_simd), 0, 0, d_out, 1.0f);   // every CU busy
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, d_out, 8, hipMemcpyDeviceToHost);
    const double n = 64.0 * 64 * 4;   // instructions per wave
    printf("%-20s %d wave(s)/SIMD: %.2f ticks per wave-instruction; kernel %.1f us -> %.3f ns per wave-instruction per SIMD\n",
           name, waves_per_simd, h / n, ms * 1e3, ms * 1e6 / (n * waves_per_simd));
}

int main()
{
    long long* d;
    hipMalloc(&d, 16);
    for (int w = 1; w <= 4; w++) {
        run<0>("v_add_f32", d, w);
        run<7>("v_mul_f32", d, w);
        run<6>("v_pk_add/mul_f32", d, w);
        run<1>("v_add_f64", d, w);
        run<5>("v_mul_f64", d, w);
        run<2>("v_fma_f64", d, w);
        run<3>("v_cvt_f64_f32", d, w);
        run<4>("v_cvt_f32_f64", d, w);
        run<8>("dep v_fma_f64", d, w);
        run<9>("dep v_add_f32", d, w);
        run<10>("dep add>mul>cvt>dadd", d, w);
    }
    return 0;
}
