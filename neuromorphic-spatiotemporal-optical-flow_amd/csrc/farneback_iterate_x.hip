// Farneback inner iteration, fused, with the box filter's ROW sums in the reference library's own order.
//
// Reference arithmetic: FarnebackUpdateMatrices + FarnebackUpdateFlow_Blur of the library behind
// cv2.calcOpticalFlowFarneback (/root/reference/optical_flow_seg.py:203).  The library forms the row sums of the
// (2m+1)^2 box as ONE running double-precision sum along each image row,
//     g += vsum[x+m] - vsum[x-m-1]        (started from vsum[0]*(m+2) + vsum[1] + .. + vsum[m-1]),
// and where the 2x2 system of a pixel is rank deficient (straight edges, flat areas) the rounding history of that sum
// decides the 4th decimal of the flow: summing each pixel's window directly (k_iterate_q, farneback_iterate.hip) leaves
// 1e-4 on the reference's real autodriving frames.  The running sum is sequential along the WHOLE row, across every
// column strip.  This kernel keeps it inside the strip walker:
//
//   * a workgroup owns a strip of SW = 192 output columns (+ 2m+1 halo columns: a ring of 208 columns) of one pair and
//     walks it top to bottom, four rows per step, like k_iterate_q: producer waves compute the rows of M into a ring in
//     LDS (per ring entry one float4 + one float: planes 0..3 move with one 16-byte LDS access);
//   * consumer waves, thread <-> OUTPUT column x, keep the running column sums of the two columns x+m and x-m-1 in
//     registers (the column sums of every column are formed twice: once as a window's entering and once as its
//     leaving column -- no exchange between threads) and publish D[x] = vsum[x+m] - vsum[x-m-1] of the step's four
//     rows to LDS;
//   * ONE scanner wave, lane <-> (row, plane) of the step, runs the library's recurrence g += D[x] over the strip's
//     columns, in place (g[x] takes D[x]'s slot);  the consumers then solve their column's four 2x2 systems;
//   * the strips of one image are chained left to right: a strip's scanner starts each row from the value its left
//     neighbour's scanner ended on (20 doubles per step), handed over through global memory as data-tagged 8-byte
//     granules {tag = launch epoch, 32 bits of payload} -- relaxed agent-scope stores and loads, no fences, no flags --
//     by a producer wave with time to spare (the scanner only sees LDS).  Strip s is therefore always a step or so
//     behind strip s-1; nothing else couples them.
//   * a workgroup takes its (pair, strip) job from a per-XCD ticket counter at start-up, in strip order, so the left
//     neighbour of a running strip is itself running or done whatever order the hardware dispatches workgroups in
//     (no assumption on dispatch order or placement; an XCD owning whole pairs is for L2 locality only).  Work lists:
//     the ticket indexes a host-built list of the (item, strip) jobs that exist, one list per XCD, an item's strips
//     consecutive in one list (nsof_launch_iterate_x_het; farneback_batch.hip builds the lists).
//
// D / g are double-buffered, so within one step (ONE workgroup barrier) the scanner runs step t while the consumers solve
// step t-1 out of the other buffer and then form and publish the column sums of step t+1 into it, and the producers
// write the rows of step t+2.  The scanner is bound by instruction issue (a dependent v_add_f64 every ~10 clocks, ~12 per
// 16-byte LDS load, ~30 per 16-byte LDS store: scripts/scan_probe.hip), about 31 clocks per column: it is the longest
// role of a step and nothing else may sit on its critical path.  Arithmetic and its order are the library's, bit for bit.
// HBM traffic per pixel per iteration: 56 B as k_iterate_q (+ 80 B per strip boundary and row of carries).
#include "iterate_common.h"

namespace {

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned gu32;
typedef __attribute__((address_space(3))) int lds_int;   // a flag word in LDS, accessed as such (ds_read / ds_write, no flat access)

// XCD (0..7) this wave runs on: HW_REG_XCC_ID (id 20), bits 3:0.
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u; }

// Geometry: 12 waves (round 4; 11 before).  Waves 0-2 consumers (192 output columns: column sums, D, and row 0 of the step's
// solves), wave 3 the scanner, waves 4-6 / 8-10 producers of rows 0,1 / 2,3 of every step for ring columns 0..191 (thread <->
// column), wave 7 the producer of all four rows for ring columns 192..207 (lane <-> (row, column)) which is also the strip's
// I/O wave and solves row 1, wave 11 the solver of rows 2,3.  Waves go to SIMD wave % 4: SIMD 3 holds the scanner, the I/O
// wave and the solver.  Why: the step is bound by vector issue, and with the solves in the consumers SIMDs 0-2 carried
// ~1144 issue units per step against ~363 on SIMD 3 (ISA mix of the three roles, scripts/isa_mix.py); a single wave issues
// one vector instruction per ~8 clocks whatever its SIMD has free, so the solves are spread over three roles by ROW --
// 256 pairs 1080p, same box: 28.12 -> 27.29 ms per step (-3 %), coarse levels -6 / -7 % (their jobs are short and the
// pipelined start-up below counts for more).
// 168 VGPRs (3 waves per SIMD), 160 KB of LDS: ring 2m+9 rows x 208 x 20 B, D / g 2 x 31 KB.  (A 4th consumer wave for 240 of
// 256 columns makes 4 waves on one SIMD: 128 VGPRs, ~190 spilled registers, 1.6x slower; and no room for the second buffer.)
template <int MH>
struct XGeom {
    static constexpr int COLS = 208, NCW = 3, RB = 4;
    static constexpr int RL = 2 * MH + 1 + 2 * RB;          // ring rows: the 2m+5 a step reads and the 4 being written
    static constexpr int HALO = 2 * MH + 1;                 // ring column of output j's entering column: j + HALO
    static constexpr int SW = NCW * 64;                     // output columns per strip
    static_assert(SW + HALO <= COLS && SW % 8 == 0, "strip geometry");
    static constexpr int NB = COLS / 64, REM = COLS % 64;   // full 64-column producer blocks, columns of the remainder wave
    static_assert(REM == 16, "the remainder wave maps 4 rows x 16 columns onto its 64 lanes");
#ifndef NSOF_X_SEG0
#define NSOF_X_SEG0 96
#endif
    static constexpr int SEG0 = NSOF_X_SEG0, SEG1 = SW - SEG0;   // the strip's two scan segments (see x_scanner_loop)
    static_assert(SEG0 % 8 == 0 && SEG1 % 8 == 0 && SEG0 >= SEG1 && SEG1 > 0, "segments: whole 8-column blocks, the left one not shorter");
    static constexpr int SVW = SW + 2;                      // doubles per (row, plane) of D / g (even: 16-byte rows)
    static constexpr int WAVES = NCW + 1 + 2 * NB + 1 + 1;   // ... + the solver wave (the last one: SIMD 3, next to the scanner)
    static constexpr int THREADS = 64 * WAVES;
#ifndef NSOF_X_CQ
#define NSOF_X_CQ 1
#endif
#ifndef NSOF_X_IOQ
#define NSOF_X_IOQ 1
#endif
    // Who solves which of a step's four rows: the consumers rows [0, CQ) (own column, before they overwrite it), the I/O
    // wave rows [CQ, CQ + IOQ), the solver wave the rest.  A wave issues one vector instruction per ~8 clocks whatever the
    // SIMD has free, so the split balances the LENGTH of the per-wave instruction streams, not only the SIMDs.
    static constexpr int CQ = NSOF_X_CQ, IOQ = NSOF_X_IOQ;
    static_assert(CQ >= 0 && IOQ >= 0 && CQ + IOQ <= 4, "rows of a step");
    static constexpr size_t SV1_BYTES = sizeof(double) * 4 * 5 * SVW;      // one buffer of D / g
    static constexpr size_t SV_BYTES = 2 * SV1_BYTES;
    static constexpr size_t VI_BYTES = sizeof(double) * 2 * 4 * 5 * MH;     // row-start column sums (strip 0), 2 buffers
    static constexpr size_t JOB_BYTES = 16;
    static constexpr size_t CB_BYTES = sizeof(double) * 2 * 2 * 20;         // row-end / row-start sums between scanner and I/O wave
    static constexpr size_t RING4_BYTES = sizeof(float) * 4 * RL * COLS, RING1_BYTES = sizeof(float) * RL * COLS;
    static constexpr size_t SMEM = SV_BYTES + VI_BYTES + JOB_BYTES + CB_BYTES + RING4_BYTES + RING1_BYTES;
};

// The ring of M rows: entry (slot, column) = planes 0..3 as one float4 + plane 4.
struct XRing {
    float4* q;
    float* c;
    int cols;
    __device__ __forceinline__ void put(int slot, int col, const float (&M)[5]) const
    {
        q[slot * cols + col] = make_float4(M[0], M[1], M[2], M[3]);
        c[slot * cols + col] = M[4];
    }
    __device__ __forceinline__ void get(int slot, int col, float (&M)[5]) const
    {
        const float4 v = q[slot * cols + col];
        M[0] = v.x; M[1] = v.y; M[2] = v.z; M[3] = v.w;
        M[4] = c[slot * cols + col];
    }
};

#ifdef NSOF_X_TIMING
// Tuning build only (scripts/build_variant.sh xt farneback_iterate_x.hip -DNSOF_X_TIMING): constant-clock time that one wave
// of each role of the workgroup with job (pair 0, strip 1) spends in each part of a step; read by scripts/x_timing.py.
__device__ unsigned long long g_xt[48];
#define XT_DECL(on_)                                                        \
    const bool xt_on = (on_);                                               \
    const bool xt_any = xt;                                                 \
    unsigned long long xt_bar = 0;                                          \
    unsigned long long xt_acc[4] = {0, 0, 0, 0}, xt_prev = __builtin_amdgcn_s_memtime()
#define XT_MARK(slot)                                                       \
    do {                                                                    \
        const unsigned long long xt_now = __builtin_amdgcn_s_memtime();     \
        xt_acc[(slot) & 3] += xt_now - xt_prev;                             \
        xt_prev = xt_now;                                                   \
    } while (0)
#define XT_BAR()                                                            \
    do {                                                                    \
        const unsigned long long xb0_ = __builtin_amdgcn_s_memtime();       \
        __syncthreads();                                                    \
        xt_bar += __builtin_amdgcn_s_memtime() - xb0_;                      \
    } while (0)
#define XT_FLUSH(base)                                                      \
    do {                                                                    \
        if (xt_on)                                                          \
            for (int k_ = 0; k_ < 4; k_++) atomicAdd(&g_xt[(base) + k_], xt_acc[k_]); \
        if (xt_any && (threadIdx.x & 63) == 0) atomicAdd(&g_xt[32 + (threadIdx.x >> 6)], xt_bar); \
    } while (0)
extern "C" int nsof_debug_xtiming(unsigned long long* out32, int reset)
{
    if (out32 && hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_xt), sizeof(g_xt)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[48] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_xt), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#else
#define XT_DECL(on_)
#define XT_MARK(slot)
#define XT_BAR() __syncthreads()
#define XT_FLUSH(base)
#endif

#ifdef NSOF_X_JOBLOG
// Tuning build only (scripts/build_variant.sh xjl farneback_iterate_x.hip -DNSOF_X_JOBLOG; scripts/x_joblog.py): per job the
// 100 MHz real-time stamps of its start, of the moment its pipeline is primed (barrier Bb) and of its end, with the CU it ran
// on -- the per-CU timeline of a launch (dispatch gaps between jobs, start-up cost, chain lag, tail).
__device__ unsigned long long g_xjob[8192 * 4];
__device__ unsigned g_xjob_n;
extern "C" int nsof_debug_xjoblog(unsigned long long* out, unsigned* n, int reset)
{
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_xjob), sizeof(g_xjob)) != hipSuccess) return -1;
    if (n && hipMemcpyFromSymbol(n, HIP_SYMBOL(g_xjob_n), sizeof(unsigned)) != hipSuccess) return -1;
    if (reset) {
        unsigned z = 0;
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_xjob_n), &z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

constexpr unsigned X_SPIN_LIMIT = 1u << 21;   // polls of a carry before giving up (seconds): the grid always drains

// ---- producers ------------------------------------------------------------------------------------------------------
// One row of M: consume the loads issued two steps ago, write the ring, issue the same row of step + 2 (i + 8) and fetch
// its flow for step + 4 (i + 16).  i = stream index = image row (clamped at the bottom).
template <int MH>
__device__ __forceinline__ void x_produce(RowIn& in, FlowSrc<false>::Raw& fl, const XRing& ring, const Planes& R0,
                                          const Planes& R1, const FlowSrc<false>& F, int W, int H, int xc, int col, int i)
{
    constexpr int RL = XGeom<MH>::RL;
    float Mn[5];
    matrix_from(in, xc, min(i, H - 1), W, H, Mn);
    ring.put((i + MH + 1) % RL, col, Mn);
    issue_row(in, R0, R1, W, H, xc, min(i + 8, H - 1), F.resolve(fl));
    fl = F.fetch(min(i + 16, H - 1));
}

// rows 0..m-1 of the window of image row 0 for this thread's column; the m+1 rows above the image replicate row 0.
// ring slot of stream index i is (i + m + 1) % RL.
template <int MH>
__device__ __forceinline__ void x_rows_above(const XRing& ring, const Planes& R0, const Planes& R1, const FlowSrc<false>& F,
                                             int W, int H, int xc, int col)
{
    // A job's start-up is a chain of memory latencies (flow -> gather address -> gather -> matrix), ~1.5-2 us each under
    // load: taken one row at a time they cost 22 us per job (3 % of a 1080p strip, 17 % of its 135-row level).  So: the
    // flow of all m rows first, then the gathers of up to four rows in flight at once.
    float2 fl[MH];
#pragma unroll
    for (int i = 0; i < MH; i++) fl[i] = F.at(min(i, H - 1));
    constexpr int GRP = 4;
#pragma unroll
    for (int g0 = 0; g0 < MH; g0 += GRP) {
        RowIn t[GRP];
#pragma unroll
        for (int k = 0; k < GRP; k++)
            if (g0 + k < MH) issue_row(t[k], R0, R1, W, H, xc, min(g0 + k, H - 1), fl[g0 + k]);
#pragma unroll
        for (int k = 0; k < GRP; k++)
            if (g0 + k < MH) {
                float Mi[5];
                matrix_from(t[k], xc, min(g0 + k, H - 1), W, H, Mi);
                if (g0 + k == 0) {
#pragma unroll
                    for (int j = 0; j <= MH + 1; j++) ring.put(j, col, Mi);   // stream indices -m-1 .. 0
                } else {
                    ring.put(g0 + k + MH + 1, col, Mi);
                }
            }
    }
}

// Barriers (all roles alike): Ba, Bb, then B(t) for t = 0 .. nimg; "window t" = B(t-1) .. B(t), window -1 = Ba .. Bb.
//   before Ba   producers: step 0 (and the rows above the image)
//   window -1   producers: step 1;   consumers: column sums of step 0
//   window t    producers: step t+2; consumers: solve, column sums of step t+1, publish D; scanner: segment 0 of step t
//               and segment 1 of step t-1; I/O wave: carries (see x_remainder_loop)
// A ring of 2m+9 rows holds exactly the rows the consumers read in a window (2m+5) and the four being written.
// Full producer wave: thread <-> ring column, rows 2 GP, 2 GP + 1 of every step.
template <int MH, int GP>
__device__ __forceinline__ void x_producer_loop(const XRing& ring, const Planes& R0, const Planes& R1,
                                                const FlowSrc<false>& F, int W, int H, int xc, int col, int nimg, bool xt)
{
    if constexpr (GP == 0) x_rows_above<MH>(ring, R0, R1, F, W, H, xc, col);
    RowIn in[2][2];
    FlowSrc<false>::Raw fl[2][2];
#pragma unroll
    for (int ts = 0; ts < 2; ts++)
#pragma unroll
        for (int rr = 0; rr < 2; rr++) {
            const int r = min(4 * ts + MH + 2 * GP + rr, H - 1);
            issue_row(in[ts][rr], R0, R1, W, H, xc, r, F.at(r));
        }
#pragma unroll
    for (int ts = 0; ts < 2; ts++)
#pragma unroll
        for (int rr = 0; rr < 2; rr++) fl[ts][rr] = F.fetch(min(4 * (ts + 2) + MH + 2 * GP + rr, H - 1));
    auto step = [&](auto tsc, int t) {
        constexpr int TS = decltype(tsc)::value;
        x_produce<MH>(in[TS][0], fl[TS][0], ring, R0, R1, F, W, H, xc, col, 4 * t + MH + 2 * GP);
        x_produce<MH>(in[TS][1], fl[TS][1], ring, R0, R1, F, W, H, xc, col, 4 * t + MH + 2 * GP + 1);
    };
#ifdef NSOF_X_PPRIO
    if (GP == 1) __builtin_amdgcn_s_setprio(NSOF_X_PPRIO);
#endif
    step(std::integral_constant<int, 0>{}, 0);
    __syncthreads();                                                                 // Ba
    step(std::integral_constant<int, 1>{}, 1);
    XT_DECL(xt && (threadIdx.x & 63) == 0);
    __syncthreads();                                                                 // Bb
    for (int t = 0; t <= nimg; t += 2) {
        XT_MARK(1);                                                                  // wait at the barrier
        step(std::integral_constant<int, 0>{}, t + 2);
        XT_MARK(0);                                                                  // rows
        XT_BAR();                                                                    // B(t)
        if (t + 1 > nimg) break;
        XT_MARK(1);
        step(std::integral_constant<int, 1>{}, t + 3);
        XT_MARK(0);
        XT_BAR();                                                                    // B(t+1)
    }
    XT_FLUSH(8 + 8 * GP);
}

// ---- the solves outside the consumers: lane <-> columns lane, lane + 64, lane + 128 of the strip ----------------------------
// In window t the scanner's g of step t-1 (left half) and of step t-2 (right half) are complete.  A wave takes the sums of its
// rows [Q0, Q0 + NQ) of all three column blocks into registers first and releases the slots (flag = t + 1), then solves the
// 2x2 systems and stores the flow: every store instruction writes 64 consecutive vectors of one image row.
template <int MH, int Q0, int NQ>
__device__ __forceinline__ void x_solve_rows(const double* sv, volatile lds_int* flag, float2* Fout, size_t fpitch, int W, int H,
                                             int x0, int nimg, double scale, int lane, int t)
{
    using G = XGeom<MH>;
    constexpr int SVW = G::SVW, NBLK = G::SW / 64;
    if constexpr (NQ > 0) {
        double g[NBLK][NQ][5];
        int us[NBLK];
#pragma unroll
        for (int b = 0; b < NBLK; b++) {
            const int col = b * 64 + lane;
            const int u = t - 1 - (col >= G::SEG0 ? 1 : 0);
            us[b] = (u >= 0 && u < nimg && x0 + col < W) ? u : -1;
            const double* p = sv + (u & 1) * (G::SV1_BYTES / sizeof(double)) + col;
#pragma unroll
            for (int q = 0; q < NQ; q++)
#pragma unroll
                for (int c = 0; c < 5; c++) g[b][q][c] = p[((q + Q0) * 5 + c) * SVW];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the sums are in registers ...
        if (lane == 0) *flag = t + 1;                           // ... their slots may take the next D
#ifndef NSOF_X_ABL_SOLVE   // timing-only ablation: no solve at all
#pragma unroll
        for (int b = 0; b < NBLK; b++) {
            if (us[b] < 0) continue;
            const int x = x0 + b * 64 + lane;
#pragma unroll
            for (int q = 0; q < NQ; q++) {
                const int yo = 4 * us[b] + q + Q0;
                const double g11 = g[b][q][0] * scale, g12 = g[b][q][1] * scale, g22 = g[b][q][2] * scale;
                const double h1 = g[b][q][3] * scale, h2 = g[b][q][4] * scale;
                const double idet = nsof_recip_normal(g11 * g22 - g12 * g12 + 1e-3);
                const float ox = (float)((g11 * h2 - g12 * h1) * idet), oy = (float)((g22 * h1 - g12 * h2) * idet);
                if (yo < H) nsof_store_stream2(reinterpret_cast<float*>(Fout + (size_t)yo * fpitch + x), ox, oy);
            }
        }
#endif
    } else {
        if (lane == 0) *flag = t + 1;
    }
}

// Remainder wave: lane <-> (row r of the step, ring column 192 + c): one row per step and thread.  With time to spare, it
// is also the strip's I/O wave: it publishes the row-end sums the scanner left in LDS (cb[0][step & 1][20]) to the right
// neighbour and fetches the left neighbour's (-> cb[1][step & 1][20], a step ahead of the scanner), so that the scanner
// itself never touches global memory: the carries move as data-tagged 8-byte granules {tag = launch epoch, 32 bits of
// payload} with relaxed agent-scope stores and loads -- no fences, no flags.
// A strip settles about a step plus the hand-off latency behind its left neighbour and then never waits (a fetch that
// finds a stale tag is repeated, which delays this strip's barrier: the lag only grows until the fetches succeed).
template <int MH>
__device__ __forceinline__ void x_remainder_loop(const XRing& ring, double* cb, const Planes& R0, const Planes& R1,
                                                 const FlowSrc<false>& F, int W, int H, int xc, int col, int r, int nimg,
                                                 gu64* cin, gu64* cout, unsigned epoch, gu32* err, bool xt, const double* sv,
                                                 volatile lds_int* ioflag, float2* Fout, size_t fpitch, int x0, double scale)
{
    const int lane = threadIdx.x & 63;
    const bool io = lane < 20;
    const int l = io ? lane : 0;
    // the left neighbour's row-end sums of step s -> cb[1][s & 1]: the loads are issued first (fetch_issue), this thread's
    // row is produced meanwhile, then the tags are checked (fetch_finish; a stale tag means the neighbour is not a full step
    // ahead yet: poll)
    // The two loads are written as asm and waited for with a COUNTED s_waitcnt: the compiler's own wait for an atomic load
    // inside this control flow was vmcnt(0), i.e. every step also waited for the row and flow loads this wave had just issued
    // for two and four steps ahead (a full memory latency, ~1800 clocks per step: harmless while the wave had nothing else to
    // do, not once it solves).  vmcnt(9): the carry loads are older than the 9 loads x_produce always issues after them
    // (8 of issue_row + the flow fetch; the asm statements clobber "memory", so none of those moves ahead of them), loads
    // return in order, so "at most 9 outstanding" means the carries have arrived.
    unsigned long long g0 = 0, g1 = 0;
    auto carry_load = [&](int s) {
        const gu64* q = cin + 40 * s + 2 * l;
        asm volatile("global_load_dwordx2 %0, %2, off sc1\n\tglobal_load_dwordx2 %1, %2, off offset:8 sc1"
                     : "=&v"(g0), "=&v"(g1) : "v"(q) : "memory");
    };
    auto fetch_issue = [&](int s) {
        if (!cin || !io) return;
        carry_load(s);
    };
    // A wait that runs out (X_SPIN_LIMIT polls, or the launch's error word already set by another strip -- looked at every
    // 256 polls) marks the launch failed and is STICKY: this strip stops polling for the rest of its walk, so a failed
    // launch drains in about one time-out, not one per step (the host then fails the call: nsof_xsync_check).
    bool dead = false;
    auto fetch_finish = [&](int s) {
        if (!cin || !io) return;
        asm volatile("s_waitcnt vmcnt(9)" : "+v"(g0), "+v"(g1) : : "memory");
        for (unsigned spins = 0; !dead; spins++) {
            if (__all((unsigned)(g0 >> 32) == epoch && (unsigned)(g1 >> 32) == epoch)) break;
            if (spins > X_SPIN_LIMIT ||
                ((spins & 255u) == 255u && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                if (lane == 0) __hip_atomic_fetch_or(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                dead = true;
                break;
            }
            __builtin_amdgcn_s_sleep(8);
            carry_load(s);
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(g0), "+v"(g1) : : "memory");
        }
        cb[40 + (s & 1) * 20 + l] = __hiloint2double((int)(unsigned)g1, (int)(unsigned)g0);
    };
    // the scanner's row-end sums of step s (cb[0][s & 1]) -> the right neighbour
    auto publish = [&](int s) {
        if (!cout || !io) return;
        const double S = cb[(s & 1) * 20 + l];
        const unsigned long long tag = (unsigned long long)epoch << 32;
        __hip_atomic_store(cout + 40 * s + 2 * l, tag | (unsigned)__double2loint(S), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(cout + 40 * s + 2 * l + 1, tag | (unsigned)__double2hiint(S), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    if (r == 0) x_rows_above<MH>(ring, R0, R1, F, W, H, xc, col);
    RowIn in[2];
    FlowSrc<false>::Raw fl[2];
#pragma unroll
    for (int ts = 0; ts < 2; ts++) {
        const int y = min(4 * ts + MH + r, H - 1);
        issue_row(in[ts], R0, R1, W, H, xc, y, F.at(y));
    }
#pragma unroll
    for (int ts = 0; ts < 2; ts++) fl[ts] = F.fetch(min(4 * (ts + 2) + MH + r, H - 1));
    x_produce<MH>(in[0], fl[0], ring, R0, R1, F, W, H, xc, col, MH + r);
    __syncthreads();                                                                 // Ba
    fetch_issue(0);
    x_produce<MH>(in[1], fl[1], ring, R0, R1, F, W, H, xc, col, 4 + MH + r);
    fetch_finish(0);
    XT_DECL(xt && lane == 0);
    __syncthreads();                                                                 // Bb
    // window t: the scanner finishes segment 1 of step t-1 (its row-end sums: published in window t+1) and starts segment
    // 0 of step t+1 in the next window (its row-start sums: fetched now)
    auto window = [&](auto tsc, int t) {
        constexpr int TS = decltype(tsc)::value;
        XT_MARK(1);
        if (t >= 2) publish(t - 2);
        if (t + 1 < nimg) fetch_issue(t + 1);
        x_produce<MH>(in[TS], fl[TS], ring, R0, R1, F, W, H, xc, col, 4 * (t + 2) + MH + r);
        XT_MARK(0);
        // its share of the step's 2x2 solves (rows [CQ, CQ + IOQ)) while the carry fetch is in flight
        x_solve_rows<MH, XGeom<MH>::CQ, XGeom<MH>::IOQ>(sv, ioflag, Fout, fpitch, W, H, x0, nimg, scale, lane, t);
        XT_MARK(3);
        if (t + 1 < nimg) fetch_finish(t + 1);
        XT_MARK(2);
        XT_BAR();                                                                    // B(t)
    };
    for (int t = 0; t <= nimg; t += 2) {
        window(std::integral_constant<int, 0>{}, t);
        if (t + 1 > nimg) break;
        window(std::integral_constant<int, 1>{}, t + 1);
    }
    publish(nimg - 1);
    x_solve_rows<MH, XGeom<MH>::CQ, XGeom<MH>::IOQ>(sv, ioflag, Fout, fpitch, W, H, x0, nimg, scale, lane, nimg + 1);   // window nimg + 1: the right half's last step
    XT_FLUSH(12);
}

// ---- consumers: thread <-> output column j of the strip ----------------------------------------------------------
// Column sums only (round 4): the 2x2 solves moved to a wave of their own on SIMD 3 (x_solver_loop) -- the instruction mix
// of a step was 1144 issue units on each of SIMDs 0-2 (a consumer + two producers) against 363 on SIMD 3 (scanner + I/O
// wave), and the step is bound by vector issue on the fuller SIMDs.  The solver reads g out of the D / g buffers at the start
// of a window and says so (sflag = window + 1); the consumers overwrite those slots with the next D at the END of their
// window, after that flag.  The right half of the strip is scanned a window later than the left one, so its threads hold a
// step's D for one more window: two register sets used alternately (the loop is unrolled by two).
template <int MH>
__device__ __forceinline__ void x_consumer_loop(const XRing& ring, double* sv, double* vinit, volatile lds_int* sflag, float2* Fout,
                                                size_t fpitch, int W, int H, int x0, int j, int nimg, double scale, bool strip0,
                                                bool xt, unsigned xjl_slot = 0)
{
    using G = XGeom<MH>;
    constexpr int RL = G::RL, SVW = G::SVW, HALO = G::HALO;
    const int rb = j, ra = j + HALO;   // ring columns of the leaving (x-m-1) and entering (x+m) image column
    // row-start sums of strip 0 (the library's "g = vsum[0]*(m+2) + vsum[1] + .. + vsum[m-1]"): vsum[0] is the leaving
    // column of output 0, vsum[k] that of output k + m + 1
    const int vik = j == 0 ? 0 : j - MH - 1;
    const bool vi_thread = strip0 && (j == 0 || (j >= MH + 2 && j <= 2 * MH));
    __syncthreads();   // Ba: the rows above the image and step 0 are in the ring
    double va[5], vb[5];
    {
        float a0[5], b0[5];
        ring.get(MH + 1, ra, a0);
        ring.get(MH + 1, rb, b0);
#pragma unroll
        for (int c = 0; c < 5; c++) {
            va[c] = (double)(a0[c] * (float)(MH + 2));   // float product, as "srow0[x]*(m+2)"
            vb[c] = (double)(b0[c] * (float)(MH + 2));
        }
#pragma unroll
        for (int i = 1; i < MH; i++) {
            ring.get(i + MH + 1, ra, a0);
            ring.get(i + MH + 1, rb, b0);
#pragma unroll
            for (int c = 0; c < 5; c++) {
                va[c] += (double)a0[c];
                vb[c] += (double)b0[c];
            }
        }
    }
    int slot_new = (2 * MH + 1) % RL;           // stream index m    -> slot 2m+1
    int slot_old = 0;                           // stream index -m-1 -> slot 0
    const int x = x0 + j;
    // The scanner works on the strip's two halves a step apart (segment 1 of step s in the window after segment 0 of step
    // s), so the threads of the right half publish their D one window late (kept in registers meanwhile): lag = 1.
    const int lag = j >= G::SEG0 ? 1 : 0;
#ifdef NSOF_X_CPRIO
    __builtin_amdgcn_s_setprio(NSOF_X_CPRIO);
#endif
    XT_DECL(xt && j == 0);
    double D[2][4][5];   // D[s & 1] = D of step s
    // step s: four more rows enter the windows of this thread's two columns -> D[P], P = s & 1
    auto column_sums = [&](auto pc, int s) {
        constexpr int P = decltype(pc)::value;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            float na[5], oa[5], nb[5], ob[5];
            ring.get(slot_new, ra, na);
            ring.get(slot_old, ra, oa);
            ring.get(slot_new, rb, nb);
            ring.get(slot_old, rb, ob);
#pragma unroll
            for (int c = 0; c < 5; c++) {
                const float da = na[c] - oa[c];
                const float db = nb[c] - ob[c];
                va[c] += (double)da;
                vb[c] += (double)db;
                D[P][q][c] = va[c] - vb[c];
            }
            if (vi_thread) {
#pragma unroll
                for (int c = 0; c < 5; c++) vinit[(((s & 1) * 4 + q) * 5 + c) * MH + vik] = vb[c];
            }
            slot_new = slot_new + 1 == RL ? 0 : slot_new + 1;
            slot_old = slot_old + 1 == RL ? 0 : slot_old + 1;
        }
    };
    // D of step p (in D[P]) -> buffer p & 1
    auto publish = [&](auto pc, int p) {
        constexpr int P = decltype(pc)::value;
        if (p < 0 || p >= nimg) return;
        double* svj = sv + (p & 1) * (G::SV1_BYTES / sizeof(double)) + j;
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int c = 0; c < 5; c++) svj[(q * 5 + c) * SVW] = D[P][q][c];
    };
    column_sums(std::integral_constant<int, 0>{}, 0);
    if (!lag) publish(std::integral_constant<int, 0>{}, 0);
    __syncthreads();   // Bb: D(0) of the left half is published, step 1 is in the ring
#ifdef NSOF_X_JOBLOG
    if (threadIdx.x == 0 && xjl_slot < 8192u) g_xjob[4 * xjl_slot + 1] = __builtin_amdgcn_s_memrealtime();
#endif
    // window t (P = t & 1): D[P] = D(t), formed a window ago; column_sums(t + 1) -> D[P ^ 1]
    auto window = [&](auto pc, int t) {
        constexpr int P = decltype(pc)::value;
        XT_MARK(1);        // wait at the barrier
        if constexpr (G::CQ > 0) {   // rows the consumers still solve themselves (own column: read before it is overwritten)
            const int u = t - 1 - lag;
            if (u >= 0 && u < nimg && x < W) {
                const double* svj = sv + (u & 1) * (G::SV1_BYTES / sizeof(double)) + j;
#pragma unroll
                for (int q = 0; q < G::CQ; q++) {
                    const int yo = 4 * u + q;
                    const double g11 = svj[(q * 5 + 0) * SVW] * scale, g12 = svj[(q * 5 + 1) * SVW] * scale;
                    const double g22 = svj[(q * 5 + 2) * SVW] * scale;
                    const double h1 = svj[(q * 5 + 3) * SVW] * scale, h2 = svj[(q * 5 + 4) * SVW] * scale;
                    const double idet = nsof_recip_normal(g11 * g22 - g12 * g12 + 1e-3);
                    const float ox = (float)((g11 * h2 - g12 * h1) * idet), oy = (float)((g22 * h1 - g12 * h2) * idet);
                    if (yo < H) nsof_store_stream2(reinterpret_cast<float*>(Fout + (size_t)yo * fpitch + x), ox, oy);
                }
            }
        }
        XT_MARK(2);        // solve
        if (t == nimg + 1) return;
        if (t + 1 < nimg) column_sums(std::integral_constant<int, P ^ 1>{}, t + 1);
        // the solver wave has taken g(t-1) (left half) / g(t-2) (right half) out of the slots D goes into now; the wait is
        // bounded like every other (a wave of this workgroup sets the flag; normally long before)
        for (int spin = 0; (sflag[0] < t + 1 || (G::IOQ > 0 && sflag[1] < t + 1)) && spin < (1 << 22); spin++)
            __builtin_amdgcn_s_sleep(1);
        if (lag) publish(std::integral_constant<int, P>{}, t);            // right half: D(t), formed a window ago
        else publish(std::integral_constant<int, P ^ 1>{}, t + 1);        // left half: D(t+1), at once
        XT_MARK(0);        // column sums
        XT_BAR();          // B(t)
    };
    for (int t = 0; t <= nimg + 1; t += 2) {
        window(std::integral_constant<int, 0>{}, t);
        if (t + 1 > nimg + 1) break;
        window(std::integral_constant<int, 1>{}, t + 1);
    }
    XT_FLUSH(0);
}

// The solver wave (the workgroup's last: SIMD 3, next to the scanner and the I/O wave).
template <int MH>
__device__ __forceinline__ void x_solver_loop(const double* sv, volatile lds_int* sflag, float2* Fout, size_t fpitch, int W, int H,
                                              int x0, int nimg, double scale, int lane, bool xt)
{
    using G = XGeom<MH>;
    __syncthreads();   // Ba
    __syncthreads();   // Bb
    XT_DECL(xt && lane == 0);
    for (int t = 0; t <= nimg + 1; t++) {
        XT_MARK(1);        // wait at the barrier
        x_solve_rows<MH, G::CQ + G::IOQ, 4 - G::CQ - G::IOQ>(sv, sflag, Fout, fpitch, W, H, x0, nimg, scale, lane, t);
        XT_MARK(2);        // read g, solve, store
        if (t <= nimg) XT_BAR();   // B(t)
    }
    XT_FLUSH(20);
}

// ---- the scanner wave -------------------------------------------------------------------------------------------------
// Bound by instruction issue (~28 clocks per column whatever the number of active lanes), so the strip is scanned as two
// segments of 96 columns by two groups of 20 lanes running the SAME instruction stream a step apart: in window t lanes 0-19
// (lane = q * 5 + c <-> row q, plane c) scan columns 0..95 of step t, lanes 20-39 columns 96..191 of step t-1, starting from
// the value the first group ended on a window earlier.
template <int MH>
__device__ __forceinline__ void x_scanner_loop(double* sv, const double* vinit, double* cb, bool has_left, int nimg, int ncols,
                                               int lane, bool xt)
{
    using G = XGeom<MH>;
    constexpr int SVW = G::SVW;
    typedef double d2 __attribute__((ext_vector_type(2)));
    const int seg = lane >= 20 ? 1 : 0;
    const int l = lane < 40 ? lane - 20 * seg : 0;
    double Smid = 0.;   // lanes 0-19: where segment 0 of the previous step ended
    __syncthreads();   // Ba
    __builtin_amdgcn_s_setprio(3);   // the one dependent chain every other wave of the step ends up waiting for
    __syncthreads();   // Bb
    XT_DECL(xt && lane == 0);
    for (int t = 0; t <= nimg; t++) {   // window t
        XT_MARK(6);    // wait at the barrier
        const int step = t - seg;
        const double Sleft = __shfl(Smid, lane >= 20 ? lane - 20 : lane);
        d2* row = reinterpret_cast<d2*>(sv + (step & 1) * (G::SV1_BYTES / sizeof(double)) + l * SVW + seg * G::SEG0);
        if (lane < 40 && step >= 0 && step < nimg) {
            double S;
            if (seg) {
                S = Sleft;
            } else if (has_left) {
                S = cb[40 + (step & 1) * 20 + l];   // fetched by the I/O wave a window ago
            } else {
                const double* vi = vinit + ((step & 1) * 20 + l) * MH;
                S = vi[0] * (double)(MH + 2);
#pragma unroll
                for (int k = 1; k < MH; k++) S += vi[k];
            }
            // The library's recurrence over the strip's columns, in place, hand-scheduled: blocks of 8 columns in two
            // register sets (v64-79 / v80-95) used alternately; a block's four 16-byte loads are issued one block ahead and
            // its four stores behind its adds; a set's last register pair carries the running sum into the other set's first
            // add, so a set is reloaded only after that add has issued.  Before a block's adds the queue then holds [its
            // loads, the previous block's stores(, the next block's loads)] and lgkmcnt(7 / 10..8) waits for exactly its own
            // loads (LDS operations of a wave complete in order).  The compiler's own schedule of this loop waited for
            // every store (its loop-header wait is the merge of two different queue states).  Reads past the strip's last
            // block land in the spare doubles / the next row and are not used.
            // blocks [b0, b0 + nb) of this lane's row
            auto scan_blocks = [&](int b0, int nb) {
                unsigned a = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(row + 4 * b0);
                int nrem = nb - 1;   // blocks after the first
                asm volatile(
                "ds_read_b128 v[64:67], %[a]\n\t"
                "ds_read_b128 v[68:71], %[a] offset:16\n\t"
                "ds_read_b128 v[72:75], %[a] offset:32\n\t"
                "ds_read_b128 v[76:79], %[a] offset:48\n\t"
                "ds_read_b128 v[80:83], %[a] offset:64\n\t"
                "ds_read_b128 v[84:87], %[a] offset:80\n\t"
                "ds_read_b128 v[88:91], %[a] offset:96\n\t"
                "ds_read_b128 v[92:95], %[a] offset:112\n\t"
                "s_waitcnt lgkmcnt(7)\n\t"
                "v_add_f64 v[64:65], %[S], v[64:65]\n\t"
                "v_add_f64 v[66:67], v[66:67], v[64:65]\n\t"
                "s_waitcnt lgkmcnt(6)\n\t"
                "v_add_f64 v[68:69], v[66:67], v[68:69]\n\t"
                "v_add_f64 v[70:71], v[70:71], v[68:69]\n\t"
                "s_waitcnt lgkmcnt(5)\n\t"
                "v_add_f64 v[72:73], v[70:71], v[72:73]\n\t"
                "v_add_f64 v[74:75], v[74:75], v[72:73]\n\t"
                "s_waitcnt lgkmcnt(4)\n\t"
                "v_add_f64 v[76:77], v[74:75], v[76:77]\n\t"
                "v_add_f64 v[78:79], v[78:79], v[76:77]\n\t"
                "s_nop 1\n\t"
                "ds_write_b128 %[a], v[64:67]\n\t"
                "ds_write_b128 %[a], v[68:71] offset:16\n\t"
                "ds_write_b128 %[a], v[72:75] offset:32\n\t"
                "ds_write_b128 %[a], v[76:79] offset:48\n\t"
                "s_cmp_lt_i32 %[n], 2\n\t"
                "s_cbranch_scc1 .Lxs_tail%=\n"
                ".Lxs_loop%=:\n\t"
                "s_waitcnt lgkmcnt(7)\n\t"
                "v_add_f64 v[80:81], v[78:79], v[80:81]\n\t"
                "ds_read_b128 v[64:67], %[a] offset:128\n\t"
                "ds_read_b128 v[68:71], %[a] offset:144\n\t"
                "ds_read_b128 v[72:75], %[a] offset:160\n\t"
                "ds_read_b128 v[76:79], %[a] offset:176\n\t"
                "v_add_f64 v[82:83], v[82:83], v[80:81]\n\t"
                "s_waitcnt lgkmcnt(10)\n\t"
                "v_add_f64 v[84:85], v[82:83], v[84:85]\n\t"
                "v_add_f64 v[86:87], v[86:87], v[84:85]\n\t"
                "s_waitcnt lgkmcnt(9)\n\t"
                "v_add_f64 v[88:89], v[86:87], v[88:89]\n\t"
                "v_add_f64 v[90:91], v[90:91], v[88:89]\n\t"
                "s_waitcnt lgkmcnt(8)\n\t"
                "v_add_f64 v[92:93], v[90:91], v[92:93]\n\t"
                "v_add_f64 v[94:95], v[94:95], v[92:93]\n\t"
                "s_nop 1\n\t"
                "ds_write_b128 %[a], v[80:83] offset:64\n\t"
                "ds_write_b128 %[a], v[84:87] offset:80\n\t"
                "ds_write_b128 %[a], v[88:91] offset:96\n\t"
                "ds_write_b128 %[a], v[92:95] offset:112\n\t"
                "s_waitcnt lgkmcnt(7)\n\t"
                "v_add_f64 v[64:65], v[94:95], v[64:65]\n\t"
                "ds_read_b128 v[80:83], %[a] offset:192\n\t"
                "ds_read_b128 v[84:87], %[a] offset:208\n\t"
                "ds_read_b128 v[88:91], %[a] offset:224\n\t"
                "ds_read_b128 v[92:95], %[a] offset:240\n\t"
                "v_add_f64 v[66:67], v[66:67], v[64:65]\n\t"
                "s_waitcnt lgkmcnt(10)\n\t"
                "v_add_f64 v[68:69], v[66:67], v[68:69]\n\t"
                "v_add_f64 v[70:71], v[70:71], v[68:69]\n\t"
                "s_waitcnt lgkmcnt(9)\n\t"
                "v_add_f64 v[72:73], v[70:71], v[72:73]\n\t"
                "v_add_f64 v[74:75], v[74:75], v[72:73]\n\t"
                "s_waitcnt lgkmcnt(8)\n\t"
                "v_add_f64 v[76:77], v[74:75], v[76:77]\n\t"
                "v_add_f64 v[78:79], v[78:79], v[76:77]\n\t"
                "s_nop 1\n\t"
                "ds_write_b128 %[a], v[64:67] offset:128\n\t"
                "ds_write_b128 %[a], v[68:71] offset:144\n\t"
                "ds_write_b128 %[a], v[72:75] offset:160\n\t"
                "ds_write_b128 %[a], v[76:79] offset:176\n\t"
                "v_add_u32 %[a], 128, %[a]\n\t"
                "s_sub_i32 %[n], %[n], 2\n\t"
                "s_cmp_ge_i32 %[n], 2\n\t"
                "s_cbranch_scc1 .Lxs_loop%=\n"
                ".Lxs_tail%=:\n\t"
                "s_cmp_lt_i32 %[n], 1\n\t"
                "s_cbranch_scc1 .Lxs_done%=\n\t"
                "s_waitcnt lgkmcnt(4)\n\t"
                "v_add_f64 v[80:81], v[78:79], v[80:81]\n\t"
                "v_add_f64 v[82:83], v[82:83], v[80:81]\n\t"
                "s_waitcnt lgkmcnt(4)\n\t"
                "v_add_f64 v[84:85], v[82:83], v[84:85]\n\t"
                "v_add_f64 v[86:87], v[86:87], v[84:85]\n\t"
                "s_waitcnt lgkmcnt(4)\n\t"
                "v_add_f64 v[88:89], v[86:87], v[88:89]\n\t"
                "v_add_f64 v[90:91], v[90:91], v[88:89]\n\t"
                "s_waitcnt lgkmcnt(4)\n\t"
                "v_add_f64 v[92:93], v[90:91], v[92:93]\n\t"
                "v_add_f64 v[94:95], v[94:95], v[92:93]\n\t"
                "s_nop 1\n\t"
                "ds_write_b128 %[a], v[80:83] offset:64\n\t"
                "ds_write_b128 %[a], v[84:87] offset:80\n\t"
                "ds_write_b128 %[a], v[88:91] offset:96\n\t"
                "ds_write_b128 %[a], v[92:95] offset:112\n\t"
                "v_mov_b64 v[78:79], v[94:95]\n"
                ".Lxs_done%=:\n\t"
                "v_mov_b64 %[S], v[78:79]\n\t"
                "s_waitcnt lgkmcnt(0)\n\t"
                    : [S] "+v"(S), [a] "+v"(a), [n] "+s"(nrem)
                    :
                    : "memory", "scc", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95");
            };
            // both segments run the stream of the shorter one together; the rest of the longer (left) one follows
            const int nb0 = ncols >> 3, nba = nb0 < G::SEG1 / 8 ? nb0 : G::SEG1 / 8;
#ifdef NSOF_X_ABL_SCAN   // timing-only ablation: the scanner runs two thirds of its blocks (what three 64-column segments would cost)
            scan_blocks(0, nba * 2 / 3);
#else
            scan_blocks(0, nba);
            if (nb0 > nba && !seg) scan_blocks(nba, nb0 - nba);
#endif
            if (seg) cb[(step & 1) * 20 + l] = S;   // the row-end sums: the I/O wave hands them to the right neighbour
            else Smid = S;
        }
        XT_MARK(5);        // scan
        XT_BAR();          // B(t)
    }
    XT_FLUSH(4);
}

// tickets: 8 counters, 32 words apart.  carry: granules, see the launcher.  n: pairs (items with HET); nstrips: strips of
// the image.  Work lists (HET) carry a job table instead: xjobs[0..7] = jobs in each XCD's list, xjobs[8 + k * nstrips + i]
// = item << 8 | strip, the i-th job of XCD k's list (nstrips = the lists' common stride); built on the host so that
// only strips that exist are ever taken and the strips of an item follow each other in one list.  The grid is exactly
// the number of jobs: a workgroup whose XCD's list has run out takes the next job of another list.
template <int MH, bool HET>
__global__ __launch_bounds__((XGeom<MH>::THREADS)) void k_iterate_x(
    const float* __restrict__ R0b, const float* __restrict__ R1b, size_t pair_stride, const float* __restrict__ flow_in,
    float* __restrict__ flow_out, int W, int H, int block_size, const nsof_het_item* __restrict__ items, int het_final,
    int n, int nstrips, unsigned long long* carry, unsigned* tickets, unsigned epoch, unsigned* err, int fault,
    const unsigned* __restrict__ xjobs)
{
    using G = XGeom<MH>;
    constexpr int SW = G::SW, COLS = G::COLS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_x[];
    double* sv = reinterpret_cast<double*>(smem_x);                                                 // [2][4 rows][5][SVW]
    double* vinit = reinterpret_cast<double*>(smem_x + G::SV_BYTES);                                // [2][4][5][MH]
    int* job = reinterpret_cast<int*>(smem_x + G::SV_BYTES + G::VI_BYTES);
    double* cb = reinterpret_cast<double*>(smem_x + G::SV_BYTES + G::VI_BYTES + G::JOB_BYTES);    // [out / in][2][20]
    XRing ring;
    ring.q = reinterpret_cast<float4*>(smem_x + G::SV_BYTES + G::VI_BYTES + G::JOB_BYTES + G::CB_BYTES);
    ring.c = reinterpret_cast<float*>(smem_x + G::SV_BYTES + G::VI_BYTES + G::JOB_BYTES + G::CB_BYTES + G::RING4_BYTES);
    ring.cols = COLS;
    const int tid = threadIdx.x;
    // ---- job: (pair, strip) from this XCD's ticket counter, strips of a pair in order
    if (tid == 0) {
        // the jobs of XCD k: pairs k, k + 8, ... strip by strip; work lists: xjobs[k] entries of list k
        const unsigned share = HET ? 0u : (unsigned)((n + 7) / 8) * (unsigned)nstrips;
        unsigned k = xcc_id();
        unsigned lim = HET ? xjobs[k] : share;
        unsigned i = __hip_atomic_fetch_add(tickets + 32 * k, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // a hand-over already timed out on this context (this launch or an earlier one of the same call): the call will
        // fail whatever this workgroup computes, so it leaves at once -- a failed call drains in one time-out
        if (__hip_atomic_load((gu32*)err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) i = 0xffffffffu, lim = 0;
        else if (i >= lim) {   // more workgroups landed on this XCD than its share: take another XCD's next job
            for (unsigned d = 1; d < 8 && i >= lim; d++) {
                k = (k + 1) & 7u;
                if constexpr (HET) lim = xjobs[k];
                i = __hip_atomic_fetch_add(tickets + 32 * k, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if constexpr (HET) {
            const unsigned e = i < lim ? xjobs[8u + k * (unsigned)nstrips + i] : 0xffffffffu;
            job[0] = e != 0xffffffffu ? (int)(e >> 8) : -1;
            job[1] = (int)(e & 255u);
        } else {
            job[0] = i < share ? (int)((i / (unsigned)nstrips) * 8u + k) : -1;
            job[1] = (int)(i % (unsigned)nstrips);
        }
        job[2] = job[3] = 0;   // "slots released up to window" flags of the solver wave and of the I/O wave
    }
    __syncthreads();
    const int pair = job[0], strip = job[1];
    if (pair < 0 || pair >= n) return;   // block-uniform
#ifdef NSOF_X_JOBLOG
    unsigned xjl_slot = 0xffffffffu;
    if (tid == 0) {
        xjl_slot = atomicAdd(&g_xjob_n, 1u);
        if (xjl_slot < 8192u) {
            g_xjob[4 * xjl_slot] = __builtin_amdgcn_s_memrealtime();
            const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));   // HW_REG_HW_ID
            g_xjob[4 * xjl_slot + 3] = ((unsigned long long)xcc_id() << 48) | ((unsigned long long)hw << 16) |
                                       ((unsigned long long)(pair & 0xfff) << 4) | (unsigned)(strip & 0xf);
        }
    }
#endif
    size_t fpitch = (size_t)W;
    gu64* cbase;
    if constexpr (HET) {
        const nsof_het_item& it = items[pair];
        W = it.wk;
        H = it.hk;
        if (strip * SW >= W) return;   // block-uniform, before any further barrier
        R0b += it.offR;
        R1b = R0b + 5 * (size_t)W * H;
        pair_stride = 0;
        flow_in += 2 * it.offF;
        cbase = (gu64*)(carry + it.offR / 2);   // an item's carries live in a mirror of its expansion block (always smaller)
        if (het_final) {
            flow_out = it.out;
            fpitch = (size_t)it.out_pitch;
        } else {
            flow_out += 2 * it.offF;
            fpitch = (size_t)W;
        }
    } else {
        if (strip * SW >= W) return;
        cbase = (gu64*)(carry + (size_t)pair * nstrips * (size_t)(((H + 3) / 4) * 40));
    }
    const int nimg = (H + 3) / 4;   // steps of the image
    const int x0 = strip * SW;
    const size_t plane = (size_t)W * H;
    const size_t poff = HET ? 0 : (size_t)pair * pair_stride;
    const Planes R0 = planes_of(R0b + poff, plane);
    const Planes R1 = planes_of(R1b + poff, plane);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef NSOF_X_TIMING
    const bool xt = pair == 0 && strip == 1;
#else
    const bool xt = false;
#endif
    volatile lds_int* sflag = (volatile lds_int*)(job + 2);
    if (wave < G::NCW) {
        float2* Fout = reinterpret_cast<float2*>(flow_out) + (HET ? 0 : (size_t)pair * plane);
#ifdef NSOF_X_JOBLOG
        x_consumer_loop<MH>(ring, sv, vinit, sflag, Fout, fpitch, W, H, x0, tid, nimg, 1. / (block_size * block_size), strip == 0, xt, xjl_slot);
        if (tid == 0 && xjl_slot < 8192u) g_xjob[4 * xjl_slot + 2] = __builtin_amdgcn_s_memrealtime();
#else
        x_consumer_loop<MH>(ring, sv, vinit, sflag, Fout, fpitch, W, H, x0, tid, nimg, 1. / (block_size * block_size), strip == 0, xt);
#endif
    } else if (wave == G::WAVES - 1) {
        float2* Fout = reinterpret_cast<float2*>(flow_out) + (HET ? 0 : (size_t)pair * plane);
        x_solver_loop<MH>(sv, sflag, Fout, fpitch, W, H, x0, nimg, 1. / (block_size * block_size), tid & 63, xt);
    } else if (wave == G::NCW) {
        const int ncols = min(G::SEG0, (W - x0 + 7) & ~7);   // columns of the left segment
        x_scanner_loop<MH>(sv, vinit, cb, strip > 0, nimg, ncols, tid & 63, xt);
    } else {
        // producers: waves NCW+1 .. NCW+NB rows 0,1 of blocks 0..NB-1; wave NCW+NB+1 the remainder; then rows 2,3
        const int pw = wave - (G::NCW + 1);
        const int lane = tid & 63;
        FlowSrc<false> F;
        F.base = reinterpret_cast<const char*>(flow_in) + (HET ? 0 : (size_t)pair * plane * 8);
        F.W = (unsigned)W;
        // wave -> role.  Waves go to SIMD (wave % 4).  NSOF_X_WAVEMAP=1: the light I/O wave shares SIMD 1 with the middle
        // consumer wave (the one that publishes both halves, the longest of a step) and a rows-2,3 producer moves next to
        // the scanner on SIMD 3: waves 4-6 rows 0,1 of blocks 0-2; 7 rows 2,3 of block 1; 8 rows 2,3 of block 0; 9 I/O; 10
        // rows 2,3 of block 2.  Default (0): 4-6 rows 0,1; 7 I/O; 8-10 rows 2,3.
#ifndef NSOF_X_WAVEMAP
#define NSOF_X_WAVEMAP 0
#endif
        bool is_io;
        int gp, blk;
        if (NSOF_X_WAVEMAP == 1) {
            is_io = pw == 5;
            gp = pw < 3 ? 0 : 1;
            blk = pw < 3 ? pw : (pw == 3 ? 1 : (pw == 4 ? 0 : 2));
        } else {
            is_io = pw == G::NB;
            gp = pw < G::NB ? 0 : 1;
            blk = pw < G::NB ? pw : pw - G::NB - 1;
        }
        if (is_io) {
            const int col = G::NB * 64 + (lane & 15), r = lane >> 4;
            const int xc = clampi(x0 - MH - 1 + col, 0, W - 1);
            F.xc = (unsigned)xc;
            const size_t per_strip = (size_t)nimg * 40;
            gu64* cin = strip > 0 ? cbase + (size_t)(strip - 1) * per_strip : nullptr;
            // NSOF_OPT_DEBUG_FAULT bit 0 (test hook): strip 0 of item 0 keeps its carries to itself
            gu64* cout = x0 + SW < W && !((fault & 1) && pair == 0 && strip == 0) ? cbase + (size_t)strip * per_strip : nullptr;
            float2* Fout = reinterpret_cast<float2*>(flow_out) + (HET ? 0 : (size_t)pair * plane);
            x_remainder_loop<MH>(ring, cb, R0, R1, F, W, H, xc, col, r, nimg, cin, cout, epoch, (gu32*)err, xt, sv, sflag + 1, Fout, fpitch,
                                 x0, 1. / (block_size * block_size));
        } else {
            const int col = blk * 64 + lane;
            const int xc = clampi(x0 - MH - 1 + col, 0, W - 1);
            F.xc = (unsigned)xc;
            if (gp == 0)
                x_producer_loop<MH, 0>(ring, R0, R1, F, W, H, xc, col, nimg, xt && blk == 0);
            else
                x_producer_loop<MH, 1>(ring, R0, R1, F, W, H, xc, col, nimg, xt && blk == 0);
        }
    }
}

template <int MH, bool HET>
int launch_x(nsof_ctx* ctx, int n, int max_w, int max_h, const float* R0, const float* R1, size_t pair_stride,
             const float* flow_in, float* flow_out, int W, int H, int winsize, const nsof_het_item* items, bool final,
             const unsigned* xjobs, int stride, int njobs)
{
    using G = XGeom<MH>;
    static_assert(G::SW == NSOF_X_STRIP, "the host's job tables count strips of NSOF_X_STRIP columns");
    if (int rc = lds_opt_in(ctx, k_iterate_x<MH, HET>, G::SMEM)) return rc;
    const int nstrips = HET ? stride : (max_w + G::SW - 1) / G::SW;
    unsigned long long* carry = nullptr;
    unsigned* tickets = nullptr;
    unsigned* err = nullptr;
    // uniform batches: [pair][strip][row][5] doubles as two granules each; work lists: the mirror of the expansion buffer
    const size_t cbytes = HET ? 0 : (size_t)n * nstrips * (size_t)((max_h + 3) / 4) * 40 * 8;
    if (int rc = nsof_xsync_reserve(ctx, cbytes, &carry, &tickets, &err)) return rc;
    NSOF_HIP(ctx, hipMemsetAsync(tickets, 0, 8 * 32 * sizeof(unsigned), ctx->stream));
    unsigned epoch = ++ctx->x_epoch;
    if (epoch == 0) {   // wrapped: every tag in the buffer is stale but may match again -> clear it
        NSOF_HIP(ctx, hipMemsetAsync(ctx->x_carry, 0, ctx->x_carry_bytes, ctx->stream));
        epoch = ++ctx->x_epoch;
    }
    const unsigned grid = HET ? (unsigned)njobs : 8u * (unsigned)((n + 7) / 8) * (unsigned)nstrips;
    hipLaunchKernelGGL((k_iterate_x<MH, HET>), dim3(grid), dim3(G::THREADS), G::SMEM, ctx->stream, R0, R1, pair_stride, flow_in,
                       flow_out, W, H, winsize, items, final ? 1 : 0, n, nstrips, carry, tickets, epoch, err, ctx->dbg_fault,
                       xjobs);
    return NSOF_OK;
}

}  // namespace

bool nsof_iterate_x_supported(int winsize, int W, int H)
{
    const int m = winsize / 2;
    return m >= 1 && m <= 7 && W >= 2 && H >= 2;
}

#define NSOF_X_SWITCH(HETV, ...)                                                                     \
    switch (winsize / 2) {                                                                           \
        case 1: rc = launch_x<1, HETV>(__VA_ARGS__); break;                                          \
        case 2: rc = launch_x<2, HETV>(__VA_ARGS__); break;                                          \
        case 3: rc = launch_x<3, HETV>(__VA_ARGS__); break;                                          \
        case 4: rc = launch_x<4, HETV>(__VA_ARGS__); break;                                          \
        case 5: rc = launch_x<5, HETV>(__VA_ARGS__); break;                                          \
        case 6: rc = launch_x<6, HETV>(__VA_ARGS__); break;                                          \
        case 7: rc = launch_x<7, HETV>(__VA_ARGS__); break;                                          \
        default: return nsof_set_error(ctx, NSOF_EUNSUPPORTED, "exact-order fused iteration supports winsize 2..15"); \
    }

// Exact-order fused iteration (the library's running row sums).  flow_in != flow_out.
int nsof_launch_iterate_x(nsof_ctx* ctx, int n_pairs, const float* R0, const float* R1, size_t pair_stride,
                          const float* flow_in, float* flow_out, int W, int H, int winsize)
{
    nsof_prof_scope ps(ctx, NSOF_K_ITERATE);
    int rc;
    NSOF_X_SWITCH(false, ctx, n_pairs, W, H, R0, R1, pair_stride, flow_in, flow_out, W, H, winsize, nullptr, false, nullptr, 0, 0)
    if (rc) return rc;
    NSOF_HIP(ctx, hipGetLastError());
    return NSOF_OK;
}

// Work-list twin; carries go to ctx's mirror of the level's expansion buffer (R_floats = its size in floats).
// d_xjobs: the job table (see the kernel): 8 counts, then 8 lists `stride` entries apart; njobs = the sum of the counts.
int nsof_launch_iterate_x_het(nsof_ctx* ctx, int n_items, const nsof_het_item* d_items, int max_w, int max_h, const float* R,
                              size_t R_floats, const float* flow_in, float* flow_out, bool final, int winsize,
                              const unsigned* d_xjobs, int stride, int njobs)
{
    if (njobs <= 0) return NSOF_OK;
    nsof_prof_scope ps(ctx, NSOF_K_ITERATE);
    unsigned long long* carry;
    unsigned *tk, *er;
    if (int rc0 = nsof_xsync_reserve(ctx, R_floats * 4, &carry, &tk, &er)) return rc0;
    int rc;
    NSOF_X_SWITCH(true, ctx, n_items, max_w, max_h, R, R, (size_t)0, flow_in, flow_out, 0, 0, winsize, d_items, final, d_xjobs,
                  stride, njobs)
    if (rc) return rc;
    NSOF_HIP(ctx, hipGetLastError());
    return NSOF_OK;
}
