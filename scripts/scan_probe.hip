// Micro-probe for the row-scan chain of k_iterate_x (csrc/farneback_iterate_x.hip): one wave per workgroup runs the in-place
// recurrence over 192 columns x 20 lanes of doubles in LDS, in several ablated forms; prints shader clocks per column.
//   hipcc --offload-arch=gfx950 -O3 -o scan_probe scan_probe.hip && ./scan_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int SVW = 194, NCOLS = 192, REPS = 200;

// MODE 0: adds only (dependent chain, no LDS)   1: loads + adds   2: adds + stores   3: full (product loop)
// MODE 4: full with 4-column blocks, loads two blocks ahead
template <int MODE>
__global__ __launch_bounds__(64) void k_probe(double* out, unsigned long long* clk, int nlanes)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* sv = reinterpret_cast<double*>(smem);
    const int lane = threadIdx.x;
    for (int i = lane; i < 20 * SVW + 64; i += 64) sv[i] = 1.0 + 1e-9 * i;
    __syncthreads();
    double S = 0.5;
    const bool act = lane < nlanes;
    const int l = act ? lane % 20 : 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (act && MODE < 5) {
        for (int rep = 0; rep < REPS; rep++) {
            unsigned a = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(sv + l * SVW);
            int nrem = NCOLS / 8 - 1;
            if constexpr (MODE == 0) {
                asm volatile(
                    "s_mov_b32 s40, %[n]\n\t"
                    ".Lp0_%=:\n\t"
                    "v_add_f64 %[S], %[S], %[S]\n\t" "v_add_f64 %[S], %[S], %[S]\n\t" "v_add_f64 %[S], %[S], %[S]\n\t" "v_add_f64 %[S], %[S], %[S]\n\t"
                    "v_add_f64 %[S], %[S], %[S]\n\t" "v_add_f64 %[S], %[S], %[S]\n\t" "v_add_f64 %[S], %[S], %[S]\n\t" "v_add_f64 %[S], %[S], %[S]\n\t"
                    "s_sub_i32 s40, s40, 1\n\t" "s_cmp_ge_i32 s40, 0\n\t" "s_cbranch_scc1 .Lp0_%=\n\t"
                    : [S] "+v"(S) : [n] "s"(nrem) : "scc", "s40");
            } else if constexpr (MODE == 1 || MODE == 2 || MODE == 3) {
#define RD(d, o) "ds_read_b128 v[" #d "], %[a] offset:" #o "\n\t"
#define WR(d, o) "ds_write_b128 %[a], v[" #d "] offset:" #o "\n\t"
                asm volatile(
                    "s_mov_b32 s40, %[n]\n\t"
                    "ds_read_b128 v[64:67], %[a]\n\t" RD(68:71, 16) RD(72:75, 32) RD(76:79, 48)
                    "v_mov_b64 v[62:63], %[S]\n\t"
                    ".Lp1_%=:\n\t"
                    "s_waitcnt lgkmcnt(%[w0])\n\t"
                    "v_add_f64 v[64:65], v[62:63], v[64:65]\n\t" "v_add_f64 v[66:67], v[66:67], v[64:65]\n\t"
                    "s_waitcnt lgkmcnt(%[w1])\n\t"
                    "v_add_f64 v[68:69], v[66:67], v[68:69]\n\t" "v_add_f64 v[70:71], v[70:71], v[68:69]\n\t"
                    "s_waitcnt lgkmcnt(%[w2])\n\t"
                    "v_add_f64 v[72:73], v[70:71], v[72:73]\n\t" "v_add_f64 v[74:75], v[74:75], v[72:73]\n\t"
                    "s_waitcnt lgkmcnt(%[w3])\n\t"
                    "v_add_f64 v[76:77], v[74:75], v[76:77]\n\t" "v_add_f64 v[62:63], v[78:79], v[76:77]\n\t"
                    "v_mov_b64 v[78:79], v[62:63]\n\t"
                    "s_nop 1\n\t"
                    ".if %[st]\n\t"
                    "ds_write_b128 %[a], v[64:67]\n\t" WR(68:71, 16) WR(72:75, 32) WR(76:79, 48)
                    ".endif\n\t"
                    "v_add_u32 %[a], 64, %[a]\n\t"
                    ".if %[ld]\n\t"
                    "ds_read_b128 v[64:67], %[a]\n\t" RD(68:71, 16) RD(72:75, 32) RD(76:79, 48)
                    ".endif\n\t"
                    "s_sub_i32 s40, s40, 1\n\t" "s_cmp_ge_i32 s40, 0\n\t" "s_cbranch_scc1 .Lp1_%=\n\t"
                    "v_mov_b64 %[S], v[62:63]\n\t"
                    "s_waitcnt lgkmcnt(0)\n\t"
                    : [S] "+v"(S), [a] "+v"(a)
                    : [n] "s"(nrem), [st] "i"(MODE != 1), [ld] "i"(MODE != 2), [w0] "i"(MODE == 2 ? 15 : 3), [w1] "i"(MODE == 2 ? 15 : 2),
                      [w2] "i"(MODE == 2 ? 15 : 1), [w3] "i"(MODE == 2 ? 15 : 0)
                    : "memory", "scc", "s40", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74",
                      "v75", "v76", "v77", "v78", "v79");
            }
        }
    }
    if (act && MODE >= 5) {
        for (int rep = 0; rep < REPS; rep++) {
            unsigned a = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(sv + l * SVW);
            int nrem = NCOLS / 8 - 1;
            // 8 dependent adds, then the block's stores in one of several flavours (no loads)
            asm volatile(
                "s_mov_b32 s40, %[n]\n\t"
                "s_mov_b64 s[42:43], exec\n\t"
                "v_mov_b64 v[62:63], %[S]\n\t"
                ".Lp5_%=:\n\t"
                ".if %[fullexec]\n\t" "s_mov_b64 exec, -1\n\t" ".endif\n\t"
                "v_add_f64 v[64:65], v[62:63], v[64:65]\n\t" "v_add_f64 v[66:67], v[66:67], v[64:65]\n\t"
                "v_add_f64 v[68:69], v[66:67], v[68:69]\n\t" "v_add_f64 v[70:71], v[70:71], v[68:69]\n\t"
                "v_add_f64 v[72:73], v[70:71], v[72:73]\n\t" "v_add_f64 v[74:75], v[74:75], v[72:73]\n\t"
                "v_add_f64 v[76:77], v[74:75], v[76:77]\n\t" "v_add_f64 v[62:63], v[78:79], v[76:77]\n\t"
                ".if %[fullexec]\n\t" "s_mov_b64 exec, s[42:43]\n\t" ".endif\n\t"
                "s_nop 1\n\t"
                ".if %[fl] == 0\n\t"
                "ds_write_b128 %[a], v[64:67]\n\t" "ds_write_b128 %[a], v[68:71] offset:16\n\t" "ds_write_b128 %[a], v[72:75] offset:32\n\t" "ds_write_b128 %[a], v[76:79] offset:48\n\t"
                ".endif\n\t"
                ".if %[fl] == 1\n\t"
                "ds_write_b64 %[a], v[64:65]\n\t" "ds_write_b64 %[a], v[66:67] offset:8\n\t" "ds_write_b64 %[a], v[68:69] offset:16\n\t" "ds_write_b64 %[a], v[70:71] offset:24\n\t"
                "ds_write_b64 %[a], v[72:73] offset:32\n\t" "ds_write_b64 %[a], v[74:75] offset:40\n\t" "ds_write_b64 %[a], v[76:77] offset:48\n\t" "ds_write_b64 %[a], v[78:79] offset:56\n\t"
                ".endif\n\t"
                ".if %[fl] == 2\n\t"
                "ds_write2_b64 %[a], v[64:65], v[66:67] offset1:1\n\t" "ds_write2_b64 %[a], v[68:69], v[70:71] offset0:2 offset1:3\n\t"
                "ds_write2_b64 %[a], v[72:73], v[74:75] offset0:4 offset1:5\n\t" "ds_write2_b64 %[a], v[76:77], v[78:79] offset0:6 offset1:7\n\t"
                ".endif\n\t"
                ".if %[fl] == 3\n\t"   // half the stores (odd columns only)
                "ds_write_b128 %[a], v[64:67]\n\t" "ds_write_b128 %[a], v[72:75] offset:32\n\t"
                ".endif\n\t"
                "v_add_u32 %[a], 64, %[a]\n\t"
                "s_sub_i32 s40, s40, 1\n\t" "s_cmp_ge_i32 s40, 0\n\t" "s_cbranch_scc1 .Lp5_%=\n\t"
                "v_mov_b64 %[S], v[62:63]\n\t"
                "s_waitcnt lgkmcnt(0)\n\t"
                : [S] "+v"(S), [a] "+v"(a)
                : [n] "s"(nrem), [fl] "i"(MODE == 9 ? 4 : (MODE - 5) & 3), [fullexec] "i"(MODE >= 9)
                : "memory", "scc", "s40", "s42", "s43", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74",
                  "v75", "v76", "v77", "v78", "v79");
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) clk[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 64 + lane] = S + sv[lane];
}

template <int MODE>
void run(const char* name, int nlanes, int blocks)
{
    double* out;
    unsigned long long* clk;
    hipMalloc(&out, blocks * 64 * sizeof(double));
    hipMalloc(&clk, blocks * sizeof(unsigned long long));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k_probe<MODE>, dim3(blocks), dim3(64), 40960, 0, out, clk, nlanes);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_probe<MODE>, dim3(blocks), dim3(64), 40960, 0, out, clk, nlanes);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), clk, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double cols = (double)REPS * NCOLS;
    printf("%-28s lanes %2d blocks %4d: %7.2f memtime ticks/column, %7.2f ns/column (wall %0.3f ms)\n", name, nlanes, blocks,
           (double)h[0] / cols, ms * 1e6 / cols, ms);
    hipFree(out);
    hipFree(clk);
}

int main()
{
    for (int blocks : {1, 256}) {
        run<0>("adds only", 20, blocks);
        run<0>("adds only", 64, blocks);
        run<1>("loads + adds", 20, blocks);
        run<2>("adds + stores", 20, blocks);
        run<3>("loads + adds + stores (naive)", 20, blocks);
        run<3>("loads + adds + stores (naive)", 64, blocks);
        run<5>("adds + 4 x write_b128", 20, blocks);
        run<5>("adds + 4 x write_b128", 64, blocks);
        run<6>("adds + 8 x write_b64", 20, blocks);
        run<7>("adds + 4 x write2_b64", 20, blocks);
        run<8>("adds + 2 x write_b128", 20, blocks);
        run<9>("adds (exec=-1), no stores", 20, blocks);
    }
    return 0;
}
