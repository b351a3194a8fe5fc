// Issue cost of packed-float VALU instructions on gfx950 against the scalar-float instructions they replace.
// One workgroup per CU, WAVES waves per SIMD, each wave runs a loop of 64 independent instructions of one kind
// (8 accumulator registers / register pairs, so dependent-issue latency is not what is measured); prints shader
// clocks per instruction per wave and per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o pk_probe pk_probe.hip && ./pk_probe
// Why: clang's SLP vectoriser packs adjacent float adds / multiplies into v_pk_add_f32 / v_pk_mul_f32; the library is
// built with -fno-slp-vectorize because the VALU-bound kernels ran 3.5 % (k_iterate_x) to 33 % (k_polyexp_rs<10>) slower
// with them.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int REPS = 2000;

#define R8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define C8(OP) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0)   // one dependent chain
#define C2(OP) OP(0) OP(1) OP(0) OP(1) OP(0) OP(1) OP(0) OP(1)   // two chains
#define PK_ADD(i) "v_pk_add_f32 %[p" #i "], %[p" #i "], %[q]\n\t"
#define PK_MUL(i) "v_pk_mul_f32 %[p" #i "], %[p" #i "], %[q]\n\t"
#define PK_FMA(i) "v_pk_fma_f32 %[p" #i "], %[p" #i "], %[q], %[q]\n\t"
#define S_ADD(i) "v_add_f32 %[s" #i "], %[s" #i "], %[c]\n\t"
#define S_MUL(i) "v_mul_f32 %[s" #i "], %[s" #i "], %[c]\n\t"
#define S_FMA(i) "v_fma_f32 %[s" #i "], %[s" #i "], %[c], %[c]\n\t"
#define D_ADD(i) "v_add_f64 %[d" #i "], %[d" #i "], %[e]\n\t"
#define D_FMA(i) "v_fma_f64 %[d" #i "], %[d" #i "], %[e], %[e]\n\t"
#define D_CVT(i) "v_cvt_f64_f32 %[d" #i "], %[s" #i "]\n\t"

typedef float v2f __attribute__((ext_vector_type(2)));

// KIND 0 v_pk_add_f32  1 v_pk_mul_f32  2 v_pk_fma_f32  3 v_add_f32  4 v_mul_f32  5 v_fma_f32  6 v_add_f64  7 v_fma_f64
//      8 v_cvt_f64_f32
template <int KIND>
__global__ void k_probe(float* out, unsigned long long* clk)
{
    v2f p0 = {1.f, 2.f}, p1 = p0, p2 = p0, p3 = p0, p4 = p0, p5 = p0, p6 = p0, p7 = p0, q = {1.0000001f, 0.9999999f};
    float s0 = 1.f + threadIdx.x, s1 = s0, s2 = s0, s3 = s0, s4 = s0, s5 = s0, s6 = s0, s7 = s0, c = 1.0000001f;
    double d0 = 1. + threadIdx.x, d1 = d0, d2 = d0, d3 = d0, d4 = d0, d5 = d0, d6 = d0, d7 = d0, e = 1.0000001;
    const bool second = (__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) & 1) != 0;   // wave-uniform
    (void)second;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < (KIND < 9 ? REPS : 0); r++) {
#define BODY(OP) BODYR(R8, OP)
#define BODYR(RP, OP)                                                                                                  \
    asm volatile(RP(OP) RP(OP) RP(OP) RP(OP) RP(OP) RP(OP) RP(OP) RP(OP)                                                   \
                 : [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2), [p3] "+v"(p3), [p4] "+v"(p4), [p5] "+v"(p5), [p6] "+v"(p6),  \
                   [p7] "+v"(p7), [s0] "+v"(s0), [s1] "+v"(s1), [s2] "+v"(s2), [s3] "+v"(s3), [s4] "+v"(s4), [s5] "+v"(s5),  \
                   [s6] "+v"(s6), [s7] "+v"(s7), [d0] "+v"(d0), [d1] "+v"(d1), [d2] "+v"(d2), [d3] "+v"(d3), [d4] "+v"(d4),  \
                   [d5] "+v"(d5), [d6] "+v"(d6), [d7] "+v"(d7)                                                              \
                 : [q] "v"(q), [c] "v"(c), [e] "v"(e))
        if constexpr (KIND == 0) BODY(PK_ADD);
        if constexpr (KIND == 1) BODY(PK_MUL);
        if constexpr (KIND == 2) BODY(PK_FMA);
        if constexpr (KIND == 3) BODY(S_ADD);
        if constexpr (KIND == 4) BODY(S_MUL);
        if constexpr (KIND == 5) BODY(S_FMA);
        if constexpr (KIND == 6) BODY(D_ADD);
        if constexpr (KIND == 7) BODY(D_FMA);
        if constexpr (KIND == 8) BODY(D_CVT);
    }
    // dependent chains (one wave per SIMD): 14-19 one chain, 20-25 two chains
#define CHAIN(K, RP, OP) if constexpr (KIND == K) { for (int r = 0; r < REPS; r++) BODYR(RP, OP); }
    CHAIN(14, C8, S_ADD) CHAIN(15, C8, S_MUL) CHAIN(16, C8, S_FMA) CHAIN(17, C8, PK_ADD) CHAIN(18, C8, PK_MUL) CHAIN(19, C8, D_ADD)
    CHAIN(20, C2, S_ADD) CHAIN(21, C2, S_MUL) CHAIN(22, C2, S_FMA) CHAIN(23, C2, PK_ADD) CHAIN(24, C2, PK_MUL) CHAIN(25, C2, D_ADD)
    // two waves per SIMD running DIFFERENT kinds (wave w and w + 4 share SIMD w % 4): do they issue side by side?
#define MIXED(K, A, B)                                        \
    if constexpr (KIND == K) {                                \
        if (second) { for (int r = 0; r < REPS; r++) BODY(A); } \
        else { for (int r = 0; r < REPS; r++) BODY(B); }        \
    }
    MIXED(9, S_ADD, D_ADD)
    MIXED(10, PK_ADD, D_ADD)
    MIXED(11, S_FMA, D_ADD)
    MIXED(12, S_MUL, D_FMA)
    MIXED(13, S_ADD, D_CVT)
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) clk[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y + s0 + s1 + s2 + s3 + s4 + s5 +
                                                 s6 + s7 + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
}

// mixed kinds: mean clocks per instruction of the waves of each half (first: waves 0-3, second: waves 4-7)
template <int KIND>
void run_mixed(float* out, unsigned long long* clk, const char* a, const char* b, bool last)
{
    const int threads = 512, blocks = 256;
    hipLaunchKernelGGL(k_probe<KIND>, dim3(blocks), dim3(threads), 0, 0, out, clk);
    hipLaunchKernelGGL(k_probe<KIND>, dim3(blocks), dim3(threads), 0, 0, out, clk);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 8);
    hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
    double s[2] = {0, 0};
    for (size_t i = 0; i < h.size(); i++) s[(i % 8) / 4] += (double)h[i];
    printf("  {\"pair_on_one_simd\": [\"%s\", \"%s\"], \"clocks_per_instr_per_wave\": [%.2f, %.2f]}%s\n", a, b, s[0] / (h.size() / 2) / (REPS * 64.0),
           s[1] / (h.size() / 2) / (REPS * 64.0), last ? "" : ",");
}

template <int KIND>
double run(int waves_per_simd, float* out, unsigned long long* clk, double clock_ratio)
{
    const int threads = 256 * waves_per_simd, blocks = 256;
    hipLaunchKernelGGL(k_probe<KIND>, dim3(blocks), dim3(threads), 0, 0, out, clk);
    hipLaunchKernelGGL(k_probe<KIND>, dim3(blocks), dim3(threads), 0, 0, out, clk);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * threads / 64);
    hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    // s_memtime ticks at 100 MHz; clock_ratio = shader clocks per tick
    return s / h.size() * clock_ratio / (REPS * 64.0);
}

int main()
{
    float* out;
    unsigned long long* clk;
    hipMalloc(&out, 256 * 1024 * 4);
    hipMalloc(&clk, 256 * 16 * 8);
    int khz = 0;
    hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    const double ratio = 1.0;   // s_memtime counts shader clocks on gfx950 (s_memrealtime is the 100 MHz one)
    const char* names[] = {"v_pk_add_f32", "v_pk_mul_f32", "v_pk_fma_f32", "v_add_f32", "v_mul_f32", "v_fma_f32", "v_add_f64", "v_fma_f64", "v_cvt_f64_f32"};
    printf("{\"_doc\": \"shader clocks per instruction seen by ONE wave (s_memtime = shader clocks; nominal clock %d kHz), 64 independent instructions per loop trip; per SIMD = per wave / waves per SIMD\",\n \"rows\": [\n", khz);
    for (int w = 1; w <= 2; w++) {
        double r[9] = {run<0>(w, out, clk, ratio), run<1>(w, out, clk, ratio), run<2>(w, out, clk, ratio), run<3>(w, out, clk, ratio), run<4>(w, out, clk, ratio),
                       run<5>(w, out, clk, ratio), run<6>(w, out, clk, ratio), run<7>(w, out, clk, ratio), run<8>(w, out, clk, ratio)};
        for (int k = 0; k < 9; k++)
            printf("  {\"instr\": \"%s\", \"waves_per_simd\": %d, \"clocks_per_instr_per_wave\": %.2f, \"clocks_per_instr_per_simd\": %.2f}%s\n", names[k], w, r[k],
                   r[k] / w, ",");
    }
    {
        const char* cn[] = {"v_add_f32", "v_mul_f32", "v_fma_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_add_f64"};
        double one[6] = {run<14>(1, out, clk, 1), run<15>(1, out, clk, 1), run<16>(1, out, clk, 1), run<17>(1, out, clk, 1), run<18>(1, out, clk, 1), run<19>(1, out, clk, 1)};
        double two[6] = {run<20>(1, out, clk, 1), run<21>(1, out, clk, 1), run<22>(1, out, clk, 1), run<23>(1, out, clk, 1), run<24>(1, out, clk, 1), run<25>(1, out, clk, 1)};
        for (int k = 0; k < 6; k++)
            printf("  {\"instr\": \"%s\", \"one_wave_per_simd_clocks_per_instr\": {\"one_dependent_chain\": %.2f, \"two_chains\": %.2f}},\n", cn[k], one[k], two[k]);
    }
    run_mixed<9>(out, clk, "v_add_f64", "v_add_f32", false);
    run_mixed<10>(out, clk, "v_add_f64", "v_pk_add_f32", false);
    run_mixed<11>(out, clk, "v_add_f64", "v_fma_f32", false);
    run_mixed<12>(out, clk, "v_fma_f64", "v_mul_f32", false);
    run_mixed<13>(out, clk, "v_cvt_f64_f32", "v_add_f32", true);
    printf(" ]}\n");
    return 0;
}
