// Exact-order Farneback iteration for SMALL batches (a lone frame pair, a handful of ROI crops): latency first.
//
// The single-kernel form (k_iterate_x) gives a (strip, image) job to one workgroup that walks the whole image height;
// a lone 1920x1080 pair is 10 such jobs on 256 CUs and one iteration takes ~0.6 ms however idle the chip is.  The
// library's summation order is sequential along y (column sums) and along x (row sums) -- but only THOSE recurrences
// are: one double addition per row / column.  So for a batch too small to fill the chip the iteration is split into
// three kernels, each as wide as its step allows, with the intermediates in HBM (they stay in the 256 MB MALL):
//   k_lat_matrices   thread <-> pixel           FarnebackUpdateMatrices                                  M [5][h][w] f32
//   k_lat_colsum     thread <-> (column, plane) the library's running column sums, top down (8 waves in turn) V [h][5][w] f64
//   k_lat_rowscan    thread <-> (row, plane)    the library's running row sums, pipelined with the 2x2 solve of the
//                                               previous 32-column tile
// Same arithmetic, same order, same bits as k_iterate_x (upstream FarnebackUpdateFlow_Blur, optflowgf.cpp); 120 B/px of
// extra HBM traffic (M and V written and read once each), which is why large batches keep the fused kernel.  Measured: a
// lone 1920x1080 call 3.9 -> 1.2 ms host to host, cross-over with the fused kernel at 70-80 (strip, image) jobs.
#include <hip/hip_runtime.h>

#include <initializer_list>
#include <utility>

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
// whole recurrence (~10 clocks); what a lone wave cannot do is keep enough rows in flight -- a wave may have 64 memory
// operations outstanding (vmcnt), ~20 rows against ~2000 clocks of latency -- nor issue the loads, float differences,
// conversions and stores of a row in the time of one addition.  So the 8 waves of a workgroup take turns: wave k owns
// rows [32 k, 32 k + 32) of every 256-row pass; long before its turn it has loaded them and formed the 32 addends, in its
// turn it only runs the 32 additions on the running sums handed over through LDS and passes them on (a turn counter in
// LDS, no workgroup barrier: the next wave starts while this one stores its rows and requests its next ones).
constexpr int LC_WAVES = 8, LC_R = 32, LC_PASS = LC_WAVES * LC_R;
template <bool HET>
__global__ __launch_bounds__(64 * LC_WAVES) void k_lat_colsum(const float* __restrict__ M, int W, int H, int m,
                                                               double* __restrict__ V, const nsof_het_item* __restrict__ items,
                                                               unsigned* err, int fault)
{
    __shared__ double carry[64];
    __shared__ int turns_done, dead;
    if constexpr (HET) {
        const nsof_het_item& it = items[blockIdx.z];
        W = it.wk;
        H = it.hk;
        if (blockIdx.x * 64 >= W) return;   // block-uniform, before the barrier
        M += it.offR / 2;
        V += it.offR / 2;
    } else {
        M += (size_t)blockIdx.z * 5 * W * H;
        V += (size_t)blockIdx.z * 5 * W * H;
    }
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int x = blockIdx.x * 64 + lane, c = blockIdx.y;
    const bool live = x < W;                                  // lanes beyond the image load a clamped column and store nothing
    // wave-uniform bases + 32-bit byte offsets (the driver checks 40 W H < 4 GB): one VALU add per address
    const char* mb = reinterpret_cast<const char*>(M + (size_t)c * W * H);
    char* vb = reinterpret_cast<char*>(V + (size_t)c * W);   // V[(y * 5 + c) * W + x]
    const unsigned xm = 4u * (unsigned)min(x, W - 1), xv = 8u * (unsigned)x;
    const unsigned mrow = 4u * (unsigned)W, vrow = 40u * (unsigned)W;
    auto Mat = [&](int row) { return *reinterpret_cast<const float*>(mb + ((unsigned)row * mrow + xm)); };
    float a[LC_R], b[LC_R];
    auto fetch = [&](int y0) {
        if (y0 - m - 1 >= 0 && y0 + LC_R + m <= H) {          // interior rows: no clamps
#pragma unroll
            for (int j = 0; j < LC_R; j++) {
                a[j] = Mat(y0 + j + m);
                b[j] = Mat(y0 + j - m - 1);
            }
        } else {
#pragma unroll
            for (int j = 0; j < LC_R; j++) {
                a[j] = Mat(min(y0 + j + m, H - 1));
                b[j] = Mat(min(max(y0 + j - m - 1, 0), H - 1));
            }
        }
    };
    fetch(wave * LC_R);
    if (wave == 0) {
        // the rows of the first window: loaded together (one memory latency, not m of them: this sum is the start of every
        // launch's critical path, ~5 us of an 8 us launch at the coarse levels), added in the library's order
        float mv[7];
#pragma unroll
        for (int y = 0; y < 7; y++) mv[y] = Mat(min(y, H - 1));
        double vs = (double)(mv[0] * (float)(m + 2));
#pragma unroll
        for (int y = 1; y < 7; y++)
            if (y < m) vs += (double)mv[y];
        carry[lane] = vs;
        if (lane == 0) {
            turns_done = 0;
            // a hand-over already timed out on this context: the call fails whatever is computed from here on
            dead = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
        }
    }
    __syncthreads();
    if (*reinterpret_cast<volatile int*>(&dead)) return;   // block-uniform
    // NSOF_OPT_DEBUG_FAULT bit 1 (test hook): one wave of one workgroup never passes its turn on
    const bool withhold = (fault & 2) && wave == 3 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0;
    for (int p = 0;; p++) {
        const int y0 = p * LC_PASS + wave * LC_R;
        if (y0 >= H) break;
        double dd[LC_R];
#pragma unroll
        for (int j = 0; j < LC_R; j++) dd[j] = (double)(a[j] - b[j]);   // the difference is rounded to float before it is added
        if (y0 + LC_PASS < H) fetch(y0 + LC_PASS);                     // this wave's rows of the next pass: in flight early
        // wait for this wave's turn (every earlier turn belongs to a resident wave of this workgroup; bounded all the same)
        const int my = p * LC_WAVES + wave;
        // (a wait that runs out is sticky for the whole workgroup -- `dead` -- so a failed launch drains in one time-out;
        // the host fails the call: nsof_xsync_check)
        int spin = 0;
        for (; spin < (1 << 22); spin++) {
            if (__builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile int*>(&turns_done)) == my) break;
            if (__builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile int*>(&dead))) { spin = 1 << 22; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (spin == (1 << 22) && lane == 0) {
            atomicOr(err, 4u);
            *reinterpret_cast<volatile int*>(&dead) = 1;
        }
        asm volatile("" ::: "memory");
        double vs = *reinterpret_cast<volatile double*>(&carry[lane]);
        const int n = min(LC_R, H - y0);
        if (n == LC_R) {
#pragma unroll
            for (int j = 0; j < LC_R; j++) {
                vs += dd[j];
                dd[j] = vs;
            }
        } else {
#pragma unroll
            for (int j = 0; j < LC_R; j++) {
                if (j < n) vs += dd[j];
                dd[j] = vs;
            }
        }
        *reinterpret_cast<volatile double*>(&carry[lane]) = vs;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // the sums are in LDS before the counter moves
        if (lane == 0 && !withhold) *reinterpret_cast<volatile int*>(&turns_done) = my + 1;
        if (live) {
            unsigned off = (unsigned)y0 * vrow + xv;
#pragma unroll
            for (int j = 0; j < LC_R; j++) {
                if (j < n) *reinterpret_cast<double*>(vb + off) = dd[j];
                off += vrow;
            }
        }
    }
}

// The library's running row sums + the 2x2 solve, pipelined inside a workgroup that owns ROWS (4 or 8) image rows:
//   chain wave    lane <-> (plane, row): per 32-column tile, the tile's window of column sums from an LDS ring (static
//                 offsets: the ring repeats its first two chunks behind the fourth, and the replicated borders are stored
//                 as columns), 32 steps of S += V[x + m] - V[x - m - 1], S -> LDS
//   solver waves  thread <-> pixel of the PREVIOUS tile: the 2x2 solve and the flow store
//   loader waves  keep the ring fed, 4 chunks of 32 columns in flight in registers (coalesced 256-B row segments of V)
// One barrier per tile.  Arithmetic and order as k_rowscan_solve / the library, bit for bit.  Rows per workgroup: 4 while
// that gives at most one workgroup per CU (the per-CU load rate is the limit: more, smaller workgroups win), else 8.
#ifndef NSOF_LR_TW
#define NSOF_LR_TW 32
#endif
#ifndef NSOF_LR_DEPTH
#define NSOF_LR_DEPTH 4
#endif
constexpr int LR_TW = NSOF_LR_TW, LR_RING = 4 * LR_TW, LR_SLOTS = 6 * LR_TW, LR_DEPTH = NSOF_LR_DEPTH;
static_assert(LR_TW == 16 || LR_TW == 32, "tile width");
template <int ROWS>
struct LRGeom {
    static_assert(ROWS == 4 || ROWS == 8 || ROWS == 16, "rows per workgroup");
    // Row-major LDS arrays: a (plane, row)'s columns are adjacent, so the chain wave moves TWO columns per LDS instruction
    // (ds_read_b128 / ds_write_b128 instead of 79 eight-byte LDS instructions per 32-column tile: level 0 of a lone 1080p
    // call 48.6 -> 45.4 us, the coarser levels 14.5 -> 13.2 -- the tile's barrier and the loaders bound it after that).  Row strides are 16 B past a multiple of 256 B and plane strides
    // 128 B past one, so the 16-byte accesses of a wave's (plane, row) lanes fall into different bank groups.
    static constexpr int SSTR = LR_SLOTS + 2;                                                // ring: [5][ROWS][SSTR]
    static constexpr int PLANE = ROWS * SSTR + (16 - (ROWS * SSTR) % 32 + 32) % 32;
    static constexpr int JSTR = LR_TW + 2;                                                   // S: [2][5][ROWS][JSTR]
    static constexpr int SPLANE = ROWS * JSTR + (16 - (ROWS * JSTR) % 32 + 32) % 32;
    static constexpr int CHAIN = (5 * ROWS + 63) / 64 * 64, SOLVE = LR_TW * ROWS, LOAD = SOLVE;
    static constexpr int THREADS = CHAIN + SOLVE + LOAD;
    static constexpr size_t SMEM = sizeof(double) * (5 * PLANE + 2 * 5 * SPLANE);
};

#ifdef NSOF_LR_TIMING
// Tuning build only (scripts/build_variant.sh lrt farneback_iterate_lat.hip -DNSOF_LR_TIMING; scripts/lr_timing.py): where one
// workgroup of the row scan spends its time.  g_lrt: [role 0..2][work cycles, barrier-wait cycles], [6] steps, [7] kernel
// shader cycles of wave 0, [8] the same span in s_memrealtime ticks (100 MHz): [7] / [8] = the shader clock in units of 100 MHz.
__device__ unsigned long long g_lrt[16];
extern "C" int nsof_debug_lrtiming(unsigned long long* out16, int reset)
{
    if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_lrt), sizeof(g_lrt)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_lrt), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

template <class F, int... Ks>
__device__ __forceinline__ void lr_steps(F& step, int s, int T, std::integer_sequence<int, Ks...>)
{
    (void)std::initializer_list<int>{(s + Ks <= T ? (step(std::integral_constant<int, Ks>{}, s + Ks), 0) : 0)...};
}

template <int MH, bool HET, int LR_ROWS>
__global__ __launch_bounds__(LRGeom<LR_ROWS>::THREADS) void k_lat_rowscan(const double* __restrict__ V, int W, int H, int block_size,
                                                            float* __restrict__ flow, const nsof_het_item* __restrict__ items,
                                                            int het_final)
{
    using G = LRGeom<LR_ROWS>;
    constexpr int LR_SSTR = G::SSTR, LR_PLANE = G::PLANE, LR_JSTR = G::JSTR, LR_SPLANE = G::SPLANE, LR_CHAIN = G::CHAIN,
                  LR_SOLVE = G::SOLVE;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_lr[];
    double* ring = reinterpret_cast<double*>(smem_lr);
    double* St = ring + 5 * LR_PLANE;
    const int tid = threadIdx.x;
    size_t fpitch;
    float2* Fout;
    if constexpr (HET) {
        const nsof_het_item& it = items[blockIdx.z];
        W = it.wk;
        H = it.hk;
        if (blockIdx.x * LR_ROWS >= H) return;   // block-uniform, before any barrier
        V += it.offR / 2;
        if (het_final) {
            Fout = reinterpret_cast<float2*>(it.out);
            fpitch = (size_t)it.out_pitch;
        } else {
            Fout = reinterpret_cast<float2*>(flow) + it.offF;
            fpitch = (size_t)W;
        }
    } else {
        V += (size_t)blockIdx.z * 5 * W * H;
        Fout = reinterpret_cast<float2*>(flow) + (size_t)blockIdx.z * W * H;
        fpitch = (size_t)W;
    }
    const int y0 = blockIdx.x * LR_ROWS;
    const int T = (W + LR_TW - 1) / LR_TW;
    const double scale = 1. / (block_size * block_size);
    const int role = __builtin_amdgcn_readfirstlane(tid < LR_CHAIN ? 0 : tid < LR_CHAIN + LR_SOLVE ? 1 : 2);   // wave-uniform: chain, solver, loader
    // loader role: thread <-> (column cx, row lr) of a chunk, its 5 planes in turn; chunk u = columns
    // [16 u, 16 u + 16), replicated beyond the image.  Wave-uniform base + 32-bit byte offsets (40 W H < 4 GB, checked by
    // the driver).
    const int li = (tid - LR_CHAIN) & (LR_SOLVE - 1);   // index within the solver / the loader group
    const int cx = li & (LR_TW - 1), lr = (li / LR_TW) & (LR_ROWS - 1);
    const char* vb = reinterpret_cast<const char*>(V);
    const unsigned vrow0 = (unsigned)min(y0 + lr, H - 1) * 5u * (unsigned)W;
    auto chunk_src = [&](int u, int k) {
        const unsigned xcl = (unsigned)min(max(u * LR_TW + cx, 0), W - 1);
        return *reinterpret_cast<const double*>(vb + (vrow0 + (unsigned)k * (unsigned)W + xcl) * 8u);
    };
    auto chunk_put = [&](int u, int k, double v) {
        double* q = ring + k * LR_PLANE + lr * LR_SSTR + ((u * LR_TW + cx) & (LR_RING - 1));
        q[0] = v;
        if (((u * LR_TW) & (LR_RING - 1)) < 2 * LR_TW) q[LR_RING] = v;   // uniform: the chunk lies in the repeated part of the ring
    };
    double regs[LR_DEPTH][5];
    if (role == 2) {
#pragma unroll
        for (int u = -1; u <= 1; u++) {
#pragma unroll
            for (int k = 0; k < 5; k++) chunk_put(u, k, chunk_src(u, k));
        }
#pragma unroll
        for (int d = 0; d < LR_DEPTH; d++) {
#pragma unroll
            for (int k = 0; k < 5; k++) regs[d][k] = chunk_src(2 + d, k);
        }
    }
    __syncthreads();
    // chain role: thread <-> (plane cc, row cr)
    const int cc = min(tid / LR_ROWS, 4), cr = tid & (LR_ROWS - 1);
    const bool chain_on = tid < 5 * LR_ROWS;
    const double* rc = ring + cc * LR_PLANE + cr * LR_SSTR;
    double S = 0.;
    if (chain_on) {
        S = rc[0] * (MH + 2);                                 // columns 0 .. m-1 of the image sit in slots 0 .. m-1
#pragma unroll
        for (int x = 1; x < MH; x++) S += rc[x];
    }
    // solver role: pixel (sj, sr) of the tile
    const int sj = li & (LR_TW - 1), sr = li / LR_TW;
#ifdef NSOF_LR_TIMING
    const bool lt_on = blockIdx.x == 1 && blockIdx.z == 0 && W >= 1024;
    const bool lt_lead = lt_on && (tid == 0 || tid == LR_CHAIN || tid == LR_CHAIN + LR_SOLVE);
    unsigned long long lt_work = 0, lt_wait = 0, lt_prev = __builtin_amdgcn_s_memtime();
    const unsigned long long lt_t0 = lt_prev, lt_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    auto step = [&](auto kc, int s) {
        constexpr int K = decltype(kc)::value;
#ifdef NSOF_LR_TIMING
        {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            lt_wait += now - lt_prev;
            lt_prev = now;
        }
#endif
        if (role == 0) {
#if !(defined(NSOF_LR_ABL) && NSOF_LR_ABL == 2)   // timing-only build: no chain
            if (chain_on) {
                const int b0 = (s * LR_TW - 8) & (LR_RING - 1);   // window columns [TW s - 8, TW s + TW + 6] at slots b0 .. b0 + TW + 14
                typedef double lr_d2 __attribute__((ext_vector_type(2)));
                const lr_d2* wp = reinterpret_cast<const lr_d2*>(rc + b0);   // b0 is even: 16-byte aligned
                double w[LR_TW + 16];
#pragma unroll
                for (int k = (7 - MH) / 2; k <= (LR_TW + 7 + MH) / 2; k++) {
                    const lr_d2 v = wp[k];
                    w[2 * k] = v.x;
                    w[2 * k + 1] = v.y;
                }
                lr_d2* so = reinterpret_cast<lr_d2*>(St + (s & 1) * 5 * LR_SPLANE + cc * LR_SPLANE + cr * LR_JSTR);
#pragma unroll
                for (int j = 0; j < LR_TW; j += 2) {
                    lr_d2 o;
                    S += w[j + 8 + MH] - w[j + 7 - MH];
                    o.x = S;
                    S += w[j + 9 + MH] - w[j + 8 - MH];
                    o.y = S;
                    so[j / 2] = o;
                }
            }
#endif
        } else if (role == 2) {
            // keep the ring fed: chunk s + 2 was requested LR_DEPTH steps ago; request chunk s + 2 + LR_DEPTH
#if !(defined(NSOF_LR_ABL) && NSOF_LR_ABL == 3)   // timing-only build: no ring refill
#if !(defined(NSOF_LR_ABL) && NSOF_LR_ABL == 4)   // timing-only build: loads without the LDS writes
#pragma unroll
            for (int k = 0; k < 5; k++) chunk_put(s + 2, k, regs[K][k]);
#else
            if (regs[K][0] == 1.2345e300) chunk_put(s + 2, 0, regs[K][1] + regs[K][2] + regs[K][3] + regs[K][4]);
#endif
#if defined(NSOF_LR_ABL) && NSOF_LR_ABL == 5      // timing-only build: LDS writes without the loads
            if (s < 0) {
#else
            if (s + 2 + LR_DEPTH <= T + 1) {
#endif
#pragma unroll
                for (int k = 0; k < 5; k++) regs[K][k] = chunk_src(s + 2 + LR_DEPTH, k);
            }
#endif
        } else {
#if defined(NSOF_LR_ABL) && NSOF_LR_ABL == 1      // timing-only build: no solve
            if (s < 0) {
#else
            if (s > 0) {
#endif
                const int x = (s - 1) * LR_TW + sj, y = y0 + sr;
                const double* sp = St + ((s - 1) & 1) * 5 * LR_SPLANE + sr * LR_JSTR + sj;
                const double g11 = sp[0] * scale, g12 = sp[LR_SPLANE] * scale, g22 = sp[2 * LR_SPLANE] * scale;
                const double h1 = sp[3 * LR_SPLANE] * scale, h2 = sp[4 * LR_SPLANE] * scale;
                if (x < W && y < H) {
                    const double idet = nsof_recip_normal(g11 * g22 - g12 * g12 + 1e-3);
                    Fout[(size_t)y * fpitch + x] =
                        make_float2((float)((g11 * h2 - g12 * h1) * idet), (float)((g22 * h1 - g12 * h2) * idet));
                }
            }
        }
#ifdef NSOF_LR_TIMING
        {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            lt_work += now - lt_prev;
            lt_prev = now;
        }
#endif
        __syncthreads();
    };
    // step T only solves the last tile (its chain / ring work is harmless); unrolled by the prefetch depth (register sets)
    for (int s = 0; s <= T; s += LR_DEPTH) lr_steps(step, s, T, std::make_integer_sequence<int, LR_DEPTH>{});
#ifdef NSOF_LR_TIMING
    if (lt_lead) {
        atomicAdd(&g_lrt[2 * role], lt_work);
        atomicAdd(&g_lrt[2 * role + 1], lt_wait);
        if (tid == 0) {
            atomicAdd(&g_lrt[6], (unsigned long long)(T + 1));
            atomicAdd(&g_lrt[7], __builtin_amdgcn_s_memtime() - lt_t0);
            atomicAdd(&g_lrt[8], __builtin_amdgcn_s_memrealtime() - lt_r0);
        }
    }
#endif
}

template <int MH, int ROWS>
int launch_lat_rowscan(nsof_ctx* ctx, int n, int W, int H, int max_h, const double* V, int winsize, float* flow_out,
                       const nsof_het_item* items, bool final)
{
    using G = LRGeom<ROWS>;
    nsof_prof_scope ps(ctx, NSOF_K_ITERATE);
    if (items) {
        if (int rc = lds_opt_in(ctx, k_lat_rowscan<MH, true, ROWS>, G::SMEM)) return rc;
        hipLaunchKernelGGL((k_lat_rowscan<MH, true, ROWS>), dim3((max_h + ROWS - 1) / ROWS, 1, n), dim3(G::THREADS), G::SMEM,
                           ctx->stream, V, 0, 0, winsize, flow_out, items, final ? 1 : 0);
    } else {
        if (int rc = lds_opt_in(ctx, k_lat_rowscan<MH, false, ROWS>, G::SMEM)) return rc;
        hipLaunchKernelGGL((k_lat_rowscan<MH, false, ROWS>), dim3((H + ROWS - 1) / ROWS, 1, n), dim3(G::THREADS), G::SMEM,
                           ctx->stream, V, W, H, winsize, flow_out, nullptr, 0);
    }
    NSOF_HIP(ctx, hipGetLastError());
    return NSOF_OK;
}

int lat_rowscan(nsof_ctx* ctx, int n, int W, int H, int max_h, const double* V, int winsize, float* flow_out,
                const nsof_het_item* items, bool final)
{
#ifdef NSOF_AB
    static const bool old = NSOF_AB_GETENV("NSOF_LAT_ROWSCAN_OLD") != nullptr;   // A/B: the two-kernel form's row scan
    if (old)
        return items ? nsof_launch_rowscan_solve_het(ctx, n, items, max_h, V, flow_out, final, winsize)
                     : nsof_launch_rowscan_solve(ctx, n, V, W, H, winsize, flow_out);
#endif
    static const int rows_env = [] { const char* e = NSOF_AB_GETENV("NSOF_LR_ROWS"); return e ? atoi(e) : 0; }();   // A/B: 4 or 8
    const bool rows4 = rows_env ? rows_env == 4 : (long long)((max_h + 3) / 4) * n <= 256;
    switch (winsize / 2) {
#define NSOF_LR(MM)                                                                                                  \
    case MM:                                                                                                         \
        return rows4 ? launch_lat_rowscan<MM, 4>(ctx, n, W, H, max_h, V, winsize, flow_out, items, final)            \
                     : launch_lat_rowscan<MM, 8>(ctx, n, W, H, max_h, V, winsize, flow_out, items, final)
        NSOF_LR(1); NSOF_LR(2); NSOF_LR(3); NSOF_LR(4); NSOF_LR(5); NSOF_LR(6); NSOF_LR(7);
#undef NSOF_LR
    }
    return nsof_set_error(ctx, NSOF_EUNSUPPORTED, "small-batch exact iteration supports winsize 2..15");
}

}  // namespace

// M: 5 floats per pixel, V: 5 doubles per pixel (per pair; work list: at offR / 2 of each item, as the two-kernel form).
int nsof_launch_iterate_lat(nsof_ctx* ctx, int n_pairs, const float* R0, const float* R1, size_t pair_stride,
                            const float* flow_in, float* flow_out, int W, int H, int winsize, float* M, double* V)
{
    unsigned long long* carry_unused;
    unsigned *tickets_unused, *err;   // the time-out word of the bounded spin in k_lat_colsum (checked at nsof_synchronize)
    if (int rc = nsof_xsync_reserve(ctx, 0, &carry_unused, &tickets_unused, &err)) return rc;
    {
        nsof_prof_scope ps(ctx, NSOF_K_ITERATE);
        hipLaunchKernelGGL(k_lat_matrices<false>, dim3((W + 63) / 64, (H + 3) / 4, n_pairs), dim3(256), 0, ctx->stream, R0, R1,
                           pair_stride, flow_in, W, H, M, nullptr);
        hipLaunchKernelGGL(k_lat_colsum<false>, dim3((W + 63) / 64, 5, n_pairs), dim3(64 * LC_WAVES), 0, ctx->stream, (const float*)M, W, H,
                           winsize / 2, V, nullptr, err, ctx->dbg_fault);
    }
    return lat_rowscan(ctx, n_pairs, W, H, H, V, winsize, flow_out, nullptr, false);
}

int nsof_launch_iterate_lat_het(nsof_ctx* ctx, int n_items, const nsof_het_item* d_items, int max_w, int max_h, const float* R,
                                const float* flow_in, float* flow_out, bool final, int winsize, float* M, double* V)
{
    unsigned long long* carry_unused;
    unsigned *tickets_unused, *err;   // the time-out word of the bounded spin in k_lat_colsum (checked at nsof_synchronize)
    if (int rc = nsof_xsync_reserve(ctx, 0, &carry_unused, &tickets_unused, &err)) return rc;
    {
        nsof_prof_scope ps(ctx, NSOF_K_ITERATE);
        hipLaunchKernelGGL(k_lat_matrices<true>, dim3((max_w + 63) / 64, (max_h + 3) / 4, n_items), dim3(256), 0, ctx->stream, R, R,
                           (size_t)0, flow_in, 0, 0, M, d_items);
        hipLaunchKernelGGL(k_lat_colsum<true>, dim3((max_w + 63) / 64, 5, n_items), dim3(64 * LC_WAVES), 0, ctx->stream, (const float*)M, 0, 0,
                           winsize / 2, V, d_items, err, ctx->dbg_fault);
    }
    return lat_rowscan(ctx, n_items, 0, 0, max_h, V, winsize, flow_out, d_items, final);
}
