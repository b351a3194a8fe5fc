// Farneback inner iteration, fused: matrix update (warped R1) + box blur + 2x2 solve in ONE kernel.
//
// Reference arithmetic: FarnebackUpdateMatrices followed by FarnebackUpdateFlow_Blur of the library behind
// cv2.calcOpticalFlowFarneback (/root/reference/optical_flow_seg.py:203).  The unfused pair of kernels
// (k_update_matrices + k_blur_solve) moves 68 + 28 = 96 B/px per iteration through HBM because the
// 5-plane matrix M is written and read back.  Here M never leaves the CU:
//
//   * a 256-thread block owns a strip of 256 image columns (SW outputs + m halo columns per side) and
//     walks down the full image height two rows per step -- the column sums are running sums from row 0
//     (each row adds double(float(M[y+m] - M[y-m-1])): the float rounding of that difference is part of the
//     reference arithmetic), so rows must be visited in order;
//   * thread <-> column computes M for the incoming row (bilinear gather of R1 at x+flow) and keeps the
//     last 2m+2 rows of M of its column in a REGISTER ring (static slots: the row loop is unrolled over
//     one ring period), plus the 5 double column sums;
//   * the R0/R1 loads of the next step (2 rows) and the flow two steps ahead are in flight while the current
//     step computes (one wave keeps ~34 loads outstanding);
//   * column sums go to LDS (20 KB); thread <-> 2 adjacent pixels forms the row sums and solves.
//
// HBM traffic per pixel per iteration: R0 20 + R1 ~20 + flow 8 read, flow 8 written = 56 B (+ halo).
#include "iterate_common.h"

namespace {


#ifdef NSOF_AB   // tuning builds only (scripts/build_variant.sh): the walker (k_iterate) and the 2-row producer / consumer kernel
                  // (k_iterate_pc: NSOF_ITERATE=walker|pc, NSOF_ITER_SPLIT, NSOF_PC_COLS128, NSOF_FOLD_UPSAMPLE); the product library
                  // runs k_iterate_q (fast row sums) and k_iterate_x (library order)
template <int MH>
struct IterGeom {
    static constexpr int RB = 2;                               // rows per step (4 spills the register ring)
    static constexpr int RING = 2 * MH + 2;                    // rows a column sum spans + the one leaving
    static constexpr int SW = (256 - 2 * MH) & ~1;             // output columns per block (a thread owns 2 pixels)
};

template <int MH>
__global__ __launch_bounds__(256, 2) void k_iterate(const float* __restrict__ R0b, const float* __restrict__ R1b,
                                                     size_t pair_stride, const float* __restrict__ flow_in,
                                                     float* __restrict__ flow_out, int W, int H, int block_size)
{
    using G = IterGeom<MH>;
    constexpr int RING = G::RING, SW = G::SW, RB = G::RB;
    __shared__ double sv[RB][5][256];

    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * SW;
    const int xc = clampi(x0 - MH + tid, 0, W - 1);
    const size_t plane = (size_t)W * H;
    const Planes R0 = planes_of(R0b + (size_t)blockIdx.z * pair_stride, plane);
    const Planes R1 = planes_of(R1b + (size_t)blockIdx.z * pair_stride, plane);
    const char* FinB = reinterpret_cast<const char*>(flow_in) + (size_t)blockIdx.z * plane * 8;
    auto flowAt = [&](int r) {   // flow_in at (row r, this thread's column)
        return *reinterpret_cast<const float2*>(FinB + ((unsigned)r * (unsigned)W + (unsigned)xc) * 8u);
    };
    float2* Fout = reinterpret_cast<float2*>(flow_out) + (size_t)blockIdx.z * plane;
    const double scale = 1. / (block_size * block_size);
    auto rowOf = [&](int i) { return min(i, H - 1); };   // stream index -> image row (replicated bottom)

    // ---- prologue: rows 0..m-1 enter the sums; the m+1 rows above the image replicate row 0
    float ring[RING][5];
    double vs[5];
    {
        RowIn in;
        float M0[5];
        issue_row(in, R0, R1, W, H, xc, 0, flowAt(0));
        matrix_from(in, xc, 0, W, H, M0);
#pragma unroll
        for (int c = 0; c < 5; c++) {
            vs[c] = (double)(M0[c] * (float)(MH + 2));   // float product, as "srow0[x]*(m+2)"
#pragma unroll
            for (int j = 0; j < RING; j++) ring[j][c] = M0[c];   // slots of rows -m-1..-1 (and 0) hold row 0
        }
#pragma unroll
        for (int i = 1; i < MH; i++) {
            float Mi[5];
            const int r = rowOf(i);
            issue_row(in, R0, R1, W, H, xc, r, flowAt(r));
            matrix_from(in, xc, r, W, H, Mi);
#pragma unroll
            for (int c = 0; c < 5; c++) {
                vs[c] += (double)Mi[c];
                ring[i][c] = Mi[c];
            }
        }
    }
    // ---- software pipeline: R loads one step (4 rows) ahead, flow two steps ahead
    RowIn in[RB];
    float2 fnext[RB];
#pragma unroll
    for (int q = 0; q < RB; q++) {
        const int r = rowOf(MH + q);
        issue_row(in[q], R0, R1, W, H, xc, r, flowAt(r));
        fnext[q] = flowAt(rowOf(MH + RB + q));
    }

    for (int yb = 0; yb < H; yb += RING) {
#pragma unroll
        for (int s = 0; s < RING / RB; s++) {
            const int y = yb + RB * s;
            if (y >= H) break;
#pragma unroll
            for (int q = 0; q < RB; q++) {
                const int i = y + MH + q;                       // stream index of the row entering the window
                const int slot_new = (RB * s + q + MH) % RING;
                const int slot_old = (RB * s + q + 2 * RING - MH - 1) % RING;
                float Mn[5];
                matrix_from(in[q], xc, rowOf(i), W, H, Mn);
                // refill the pipeline: this slot now loads row i+RB; its flow was fetched a step ago
                issue_row(in[q], R0, R1, W, H, xc, rowOf(i + RB), fnext[q]);
                fnext[q] = flowAt(rowOf(i + 2 * RB));
#pragma unroll
                for (int c = 0; c < 5; c++) {
                    const float d = Mn[c] - ring[slot_old][c];
                    vs[c] += (double)d;
                    ring[slot_new][c] = Mn[c];
                    sv[q][c][tid] = vs[c];
                }
            }
            __syncthreads();
            const int hrow = tid >> 7, t = tid & 127;   // 128 threads per row, 2 pixels each
            const int yo = y + hrow, xo = x0 + 2 * t;
            if (2 * t < SW && yo < H && xo < W) {
                double g[5];
                float2 o[2];
#pragma unroll
                for (int p = 0; p < 2; p++) {
                    if (p == 0) {
#pragma unroll
                        for (int c = 0; c < 5; c++) {
                            double a = 0;
#pragma unroll
                            for (int j = 0; j <= 2 * MH; j++) a += sv[hrow][c][2 * t + j];
                            g[c] = a;
                        }
                    } else {
#pragma unroll
                        for (int c = 0; c < 5; c++) g[c] += sv[hrow][c][2 * t + 1 + 2 * MH] - sv[hrow][c][2 * t];
                    }
                    const double g11 = g[0] * scale, g12 = g[1] * scale, g22 = g[2] * scale;
                    const double h1 = g[3] * scale, h2 = g[4] * scale;
                    const double idet = nsof_recip_normal(g11 * g22 - g12 * g12 + 1e-3);
                    o[p].x = (float)((g11 * h2 - g12 * h1) * idet);
                    o[p].y = (float)((g22 * h1 - g12 * h2) * idet);
                }
                float2* dst = Fout + (size_t)yo * W + xo;
                dst[0] = o[0];
                if (xo + 1 < W) dst[1] = o[1];
            }
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Role-specialised variant (the one the driver uses): 768 threads = 12 waves per strip, one block per CU.
//
// The plain walker above is limited by registers (the M ring and the loads in flight compete for one
// thread's budget) and by code size (static ring slots force unrolling a whole ring period).  Here:
//   waves 0-3   consumers  thread <-> column: the 5 double column sums in registers; row sums + solve
//   waves 4-7   producers A, waves 8-11 producers B: stateless; thread <-> column computes M for row A resp.
//               B of every 2-row step and keeps the R0/R1 loads of its next FOUR steps in flight
//               -> 8 rows of loads in flight per column, and both groups work in every window.
// M rows go from producers to consumers through a ring in LDS (2m+6 rows x 5 planes x 256 columns,
// 100 KB for winsize 15) that also serves as the window history (M[y-m-1] is read back from it), indexed
// dynamically -- no unrolling.  ONE block barrier per 2-row step (the column sums are double-buffered), and the
// producers' gather for step s+2 overlaps the consumers' row sums + solve of step s.  Arithmetic and its order are identical
// to k_iterate / the unfused kernels.
// ---------------------------------------------------------------------------------------------

template <int MH, int COLS>
struct PCGeom {
    static constexpr int RL = 2 * MH + 6;                   // ring rows: window 2m+1, the leaving row, 2 being
                                                            // consumed next, 2 being produced
    static constexpr int SW = (COLS - 2 * MH) & ~1;         // output columns per block
    // Column sums of one (row, plane) in LDS.  HIER (windows of 5 columns and more): instead of the COLS sums v[i]
    // themselves, the even columns A[j] = v[2j], the pair sums P[j] = v[2j] + v[2j+1] and the quad sums
    // Q[j] = P[2j] + P[2j+1] (+ one 0.0) -- a (2m+1)-window sum then takes about m/2 + 5 LDS reads instead of
    // 2m + 3.  The row sums are bound by LDS bandwidth (every column sum used to be read 2m+1 times by 4 waves at
    // once, right after the barrier), see docs/HISTORY_r1_r3.md section 5.1.
#ifdef NSOF_NO_HIER   // A/B build: the plain layout (every column sum read 2m+1 times)
    static constexpr bool HIER = false;
#else
    static constexpr bool HIER = MH >= 2;
#endif
    static constexpr int HALF = COLS / 2, ZIDX = COLS + COLS / 4;
    static constexpr int SVW = HIER ? COLS + COLS / 4 + 2 : COLS;   // doubles per (row, plane)
    static constexpr size_t SV_BYTES = sizeof(double) * 4 * 5 * SVW;
    static constexpr size_t SMEM = SV_BYTES + sizeof(float) * RL * 5 * COLS;
};

// A double from another lane of the same quad (DPP quad_perm: full-rate VALU moves, no LDS crossbar).
template <int CTRL>
__device__ __forceinline__ double dpp_quad(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
constexpr int DPP_LANE_PLUS1 = 0xF5;   // quad_perm:[1,1,3,3]: even lanes read their right neighbour
constexpr int DPP_LANE_PLUS2 = 0xEE;   // quad_perm:[2,3,2,3]: lanes 0,1 of a quad read lanes 2,3

// Producer group G owns row G of every 2-row step: stream index i(t) = 2t + m + G.  It produces M for step t
// from the loads in slot J = t & 3, then refills that slot with step t+4 (4 rows = 4 steps of loads in flight
// per thread).  The gather addresses depend on the flow, so the flow of step t+4 was itself fetched four
// windows earlier (fl[J]); a one-window flow lookahead makes every window wait for a full memory latency.
template <int MH, int COLS, int G, int J, int NS, typename FS>
__device__ __forceinline__ void produce_row(RowIn (&in)[NS], typename FS::Raw (&fl)[NS], float (*mring)[5][COLS],
                                            const Planes& R0,
                                            const Planes& R1, const FS& F, int W, int H, int xc, int col, int t)
{
    constexpr int RL = PCGeom<MH, COLS>::RL;
    const int i = 2 * t + MH + G;
#if defined(NSOF_ABL) && (NSOF_ABL == 4 || NSOF_ABL == 5)   // timing-only build: idle producers
    return;
#endif
    float Mn[5];
    matrix_from(in[J], xc, min(i, H - 1), W, H, Mn);
    const int slot = (i + MH + 1) % RL;      // stream index -MH-1 (first replicated row) lives in slot 0
#pragma unroll
    for (int c = 0; c < 5; c++) mring[slot][c][col] = Mn[c];
    issue_row(in[J], R0, R1, W, H, xc, min(i + 2 * NS, H - 1), F.resolve(fl[J]));   // step t + NS
    fl[J] = F.fetch(min(i + 4 * NS, H - 1));                                         // flow of step t + 2 NS
}

// Producer waves of group G: their own loop, with exactly the same barrier sequence as the consumers.
// NS = steps of loads in flight per thread (4 in the 12-wave layout; 3 in the 16-wave layout, whose 4 waves per
// SIMD leave 128 VGPRs per wave).
template <int MH, int COLS, int G, int NS, typename FS>
__device__ __forceinline__ void producer_loop(float (*mring)[5][COLS], const Planes& R0, const Planes& R1,
                                              const FS& F, int W, int H, int xc, int col, int nsteps)
{
    RowIn in[NS];
    typename FS::Raw fl[NS];
#ifdef NSOF_PRODUCER_PRIO
    __builtin_amdgcn_s_setprio(NSOF_PRODUCER_PRIO);   // experiment: issue the gather ahead of the consumers' arithmetic
#endif
    auto flowAt = [&](int r) { return F.at(r); };
#pragma unroll
    for (int j = 0; j < NS; j++) {   // steps 0..NS-1 in flight
        const int r = min(2 * j + MH + G, H - 1);
        issue_row(in[j], R0, R1, W, H, xc, r, flowAt(r));
    }
#pragma unroll
    for (int j = 0; j < NS; j++) fl[j] = F.fetch(min(2 * (j + NS) + MH + G, H - 1));   // flows of steps NS..2NS-1
    // Barrier sequence (identical in all roles): B_init, then B(s) for s = 0..nsteps-1.
    //   before B_init            step 0 is produced
    //   between B_init and B(0)  step 1                              (consumers: column sums of step 0)
    //   between B(s) and B(s+1)  step s+2                            (consumers: row sums + solve of step s,
    //                                                                 column sums of step s+1)
    produce_row<MH, COLS, G, 0, NS>(in, fl, mring, R0, R1, F, W, H, xc, col, 0);
    __syncthreads();
    produce_row<MH, COLS, G, 1, NS>(in, fl, mring, R0, R1, F, W, H, xc, col, 1);
    auto one = [&](auto uc, int s) {   // step s + 2 uses slot (s + 2) % NS; s = sb + u with sb a multiple of NS
        constexpr int u = decltype(uc)::value;
        produce_row<MH, COLS, G, (u + 2) % NS, NS>(in, fl, mring, R0, R1, F, W, H, xc, col, s + 2);
    };
    for (int sb = 0; sb < nsteps; sb += NS) {
        if (sb < nsteps) { __syncthreads(); one(std::integral_constant<int, 0>{}, sb); }
        if (sb + 1 < nsteps) { __syncthreads(); one(std::integral_constant<int, 1>{}, sb + 1); }
        if (sb + 2 < nsteps) { __syncthreads(); one(std::integral_constant<int, 2>{}, sb + 2); }
        if constexpr (NS > 3) {
            if (sb + 3 < nsteps) { __syncthreads(); one(std::integral_constant<int, 3>{}, sb + 3); }
        }
    }
}

// Fout / fpitch: the output field and its row pitch in float2 units (W for the dense batch layout; the work-list
// path writes the last iteration straight into the caller's possibly strided field).
// DO_COL / DO_SOLVE: the consumer's two halves.  Both true: one wave group does them one after the other (12-wave
// layout).  In the 16-wave layout they are two wave groups working side by side: between B(s) and B(s+1) the
// column-sum group forms the sums of step s+1 (into the other sv buffer) while the solve group consumes those of
// step s -- same data flow and barrier sequence, half the dependent chain per wave and barrier interval.
template <int MH, int COLS, bool DO_COL, bool DO_SOLVE, typename FS>
__device__ __forceinline__ void consumer_loop(float (*mring)[5][COLS], void* sv_raw, const Planes& R0,
                                              const Planes& R1, const FS& F, float2* Fout, size_t fpitch, int W, int H,
                                              int x0, int xc, int col, int nsteps, double scale)
{
    using G = PCGeom<MH, COLS>;
    double (*sv)[5][G::SVW] = reinterpret_cast<double (*)[5][G::SVW]>(sv_raw);   // [2 buffers x 2 rows][5][SVW]
    constexpr int RL = G::RL, SW = G::SW, HT = COLS / 2;   // HT threads per output row, 2 pixels each
    constexpr bool HIER = G::HIER;
    constexpr int HALF = G::HALF, ZIDX = G::ZIDX;
    if (HIER && DO_COL && col < 20) sv[col / 5][col % 5][ZIDX] = 0.0;   // the "nothing to add" slot of every (row, plane)
    double vs[5];
#ifdef NSOF_CONSUMER_PRIO
    __builtin_amdgcn_s_setprio(NSOF_CONSUMER_PRIO);   // experiment: the consumers' dependent chain first
#endif
    auto flowAt = [&](int r) { return F.at(r); };
    if constexpr (DO_COL) {
        // prologue: rows 0..m-1 enter the sums; the m+1 rows above the image replicate row 0.
        // ring slot of stream index i is (i + m + 1) % RL.
        RowIn t;
        float M0[5];
        issue_row(t, R0, R1, W, H, xc, 0, flowAt(0));
        matrix_from(t, xc, 0, W, H, M0);
#pragma unroll
        for (int c = 0; c < 5; c++) {
            vs[c] = (double)(M0[c] * (float)(MH + 2));   // float product, as "srow0[x]*(m+2)"
#pragma unroll
            for (int j = 0; j <= MH + 1; j++) mring[j][c][col] = M0[c];   // stream indices -m-1 .. 0
        }
#pragma unroll
        for (int i = 1; i < MH; i++) {
            float Mi[5];
            const int r = min(i, H - 1);
            issue_row(t, R0, R1, W, H, xc, r, flowAt(r));
            matrix_from(t, xc, r, W, H, Mi);
#pragma unroll
            for (int c = 0; c < 5; c++) {
                vs[c] += (double)Mi[c];
                mring[i + MH + 1][c][col] = Mi[c];
            }
        }
    }
    __syncthreads();   // B_init: step 0 is in the ring
    const int hrow = col / HT, t = col % HT;
    int slot_new = (2 * MH + 1) % RL;           // stream index m   -> slot 2m+1
    int slot_old = 0;                           // stream index -m-1 -> slot 0
    auto column_sums = [&](int buf) {           // two more rows enter the window of this thread's column
#if defined(NSOF_ABL) && (NSOF_ABL == 4 || NSOF_ABL == 6)   // timing-only build: idle consumers
        return;
#endif
#pragma unroll
        for (int q = 0; q < 2; q++) {
            if constexpr (HIER) {
                double pr[5], qd[5];
#pragma unroll
                for (int c = 0; c < 5; c++) {
                    const float d = mring[slot_new][c][col] - mring[slot_old][c][col];
                    vs[c] += (double)d;
                    pr[c] = vs[c] + dpp_quad<DPP_LANE_PLUS1>(vs[c]);   // even lanes: v[col] + v[col+1]
                    qd[c] = pr[c] + dpp_quad<DPP_LANE_PLUS2>(pr[c]);   // lanes 0 mod 4: v[col..col+3]
                }
                if ((col & 1) == 0) {
#pragma unroll
                    for (int c = 0; c < 5; c++) {
                        sv[2 * buf + q][c][col >> 1] = vs[c];
                        sv[2 * buf + q][c][HALF + (col >> 1)] = pr[c];
                    }
                }
                if ((col & 3) == 0) {
#pragma unroll
                    for (int c = 0; c < 5; c++) sv[2 * buf + q][c][2 * HALF + (col >> 2)] = qd[c];
                }
            } else {
#pragma unroll
                for (int c = 0; c < 5; c++) {
                    const float d = mring[slot_new][c][col] - mring[slot_old][c][col];
                    vs[c] += (double)d;
                    sv[2 * buf + q][c][col] = vs[c];
                }
            }
            slot_new = slot_new + 1 == RL ? 0 : slot_new + 1;
            slot_old = slot_old + 1 == RL ? 0 : slot_old + 1;
        }
    };
    if constexpr (DO_COL) column_sums(0);
    for (int s = 0; s < nsteps; s++) {
        __syncthreads();   // B(s): column sums of step s visible; rows of step s+1 are in the ring
        const int buf = s & 1;
        const int yo = 2 * s + hrow, xo = x0 + 2 * t;
        if constexpr (!DO_SOLVE) {
            (void)xo; (void)yo;
            if (s + 1 < nsteps) column_sums(buf ^ 1);
            continue;
        }
#if defined(NSOF_ABL) && (NSOF_ABL == 4 || NSOF_ABL == 6)
        (void)xo; (void)yo;
#elif defined(NSOF_ABL) && NSOF_ABL == 2   // timing-only build: no row sums / solve
        if (2 * t < SW && yo < H && xo < W) {
            float2* dst = Fout + (size_t)yo * fpitch + xo;
            dst[0] = make_float2((float)sv[2 * buf + hrow][0][2 * t], (float)sv[2 * buf + hrow][1][2 * t]);
            if (xo + 1 < W) dst[1] = make_float2((float)sv[2 * buf + hrow][2][2 * t + 1], (float)sv[2 * buf + hrow][3][2 * t + 1]);
        }
#else
        if (2 * t < SW && yo < H && xo < W) {
            const double (*svr)[G::SVW] = sv[2 * buf + hrow];
            double g[5], g1[5];
            float2 o[2];
            if constexpr (HIER) {
                // window of pixel 2t: columns 2t .. 2t+2m = pairs t .. t+m-1 and the even column 2(t+m);
                // pixel 2t+1: the same minus column 2t plus column 2t+2m+1 (= P[t+m] - A[t+m]).
                // The m pairs: an odd first pair and / or an odd last pair alone, quads in between.
                const int s0 = t + (t & 1), r = t + MH - s0, nq = r >> 1;
                constexpr int NQMAX = MH / 2;
#pragma unroll
                for (int c = 0; c < 5; c++) {
                    double a;
                    if constexpr (MH & 1) {   // exactly one lone pair: the first (t odd) or the last (t even)
                        a = svr[c][HALF + ((t & 1) ? t : t + MH - 1)];
                    } else {                  // none (t even) or both (t odd)
                        a = svr[c][(t & 1) ? HALF + t : ZIDX] + svr[c][(t & 1) ? HALF + t + MH - 1 : ZIDX];
                    }
#pragma unroll
                    for (int q = 0; q < NQMAX; q++) a += svr[c][(q < nq) ? 2 * HALF + (s0 >> 1) + q : ZIDX];
                    const double am = svr[c][t + MH];
                    g[c] = a + am;
                    g1[c] = g[c] - svr[c][t] + (svr[c][HALF + t + MH] - am);
                }
            }
#pragma unroll
            for (int p = 0; p < 2; p++) {
                if constexpr (HIER) {
                    if (p == 1) {
#pragma unroll
                        for (int c = 0; c < 5; c++) g[c] = g1[c];
                    }
                } else if (p == 0) {
#pragma unroll
                    for (int c = 0; c < 5; c++) {
                        double a = 0;
#pragma unroll
                        for (int j = 0; j <= 2 * MH; j++) a += svr[c][2 * t + j];
                        g[c] = a;
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < 5; c++) g[c] += svr[c][2 * t + 1 + 2 * MH] - svr[c][2 * t];
                }
                const double g11 = g[0] * scale, g12 = g[1] * scale, g22 = g[2] * scale;
                const double h1 = g[3] * scale, h2 = g[4] * scale;
                const double idet = nsof_recip_normal(g11 * g22 - g12 * g12 + 1e-3);
                o[p].x = (float)((g11 * h2 - g12 * h1) * idet);
                o[p].y = (float)((g22 * h1 - g12 * h2) * idet);
            }
            float2* dst = Fout + (size_t)yo * fpitch + xo;
            if (xo + 1 < W && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {   // dense layout, even W: always
                nsof_store_stream4(reinterpret_cast<float*>(dst), o[0].x, o[0].y, o[1].x, o[1].y);
            } else {
                dst[0] = o[0];
                if (xo + 1 < W) dst[1] = o[1];
            }
        }
#endif
        // column sums of the next step go to the other buffer: no barrier needed in between
        if constexpr (DO_COL)
            if (s + 1 < nsteps) column_sums(buf ^ 1);
    }
}

// UPS: flow_in is the coarser level's flow [sh][sw][2] (ups_* describe it) instead of this level's own buffer.
struct UpsArgs {
    int sw, sh;
    double scale_x, scale_y;
    float mul;
};
// HET: work-list launch -- blockIdx.z indexes a device table of items of different shapes (nsof_het_item): R0b is
// the level's expansion buffer, flow_in / flow_out the level's flow buffers (item fields at offF), and with
// het_final the flow goes to the item's own output field instead.
// SPLIT: 16 waves per strip (solve | column sums | producers A | producers B) instead of 12.
template <int MH, int COLS, bool UPS, bool HET = false, bool SPLIT = false>
__global__ __launch_bounds__((SPLIT ? 4 : 3) * COLS) void k_iterate_pc(const float* __restrict__ R0b, const float* __restrict__ R1b,
                                                     size_t pair_stride, const float* __restrict__ flow_in,
                                                     float* __restrict__ flow_out, int W, int H, int block_size,
                                                     UpsArgs ups, const nsof_het_item* __restrict__ items = nullptr,
                                                     int het_final = 0)
{
    constexpr int SW = PCGeom<MH, COLS>::SW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_pc[];
    using PG = PCGeom<MH, COLS>;
    void* sv = smem_pc;                                                              // [2 buffers x 2 rows][5][SVW] doubles
    float (*mring)[5][COLS] = reinterpret_cast<float (*)[5][COLS]>(smem_pc + PG::SV_BYTES);  // [RL]

    const int tid = threadIdx.x, col = tid % COLS;
    const int role = __builtin_amdgcn_readfirstlane(tid / COLS);   // wave-uniform: 0 consumer, 1/2 producers
    // XCD-aware placement: workgroups are dealt round-robin to the 8 XCDs (linear id % 8), each with its own L2.
    // With the natural (strip, pair) order the 8 strips of a pair land on 8 different L2s and their shared halo
    // columns and gather rows are fetched once per XCD; remapped, an XCD owns whole pairs.
    int strip = blockIdx.x, pair = blockIdx.z;
    size_t fpitch = 0;
    if constexpr (HET) {
        static_assert(!UPS, "the work-list path resamples the flow with its own launch");
        const nsof_het_item& it = items[blockIdx.z];
        W = it.wk;
        H = it.hk;
        if (blockIdx.x * SW >= W) return;   // block-uniform, before any barrier
        const size_t nk = (size_t)W * H;
        R0b += it.offR;
        R1b = R0b + 5 * nk;
        pair_stride = 0;
        pair = 0;
        flow_in += 2 * it.offF;
        if (het_final) {
            flow_out = it.out;
            fpitch = (size_t)it.out_pitch;
        } else {
            flow_out += 2 * it.offF;
            fpitch = (size_t)W;
        }
    } else {
        fpitch = (size_t)W;
    }
#ifndef NSOF_NO_XCD_REMAP
    if constexpr (!HET) {
        const unsigned total = gridDim.x * gridDim.z;
        if ((total & 7u) == 0) {
            const unsigned lin = blockIdx.x + gridDim.x * blockIdx.z;
            const unsigned j = (lin & 7u) * (total >> 3) + (lin >> 3);
            pair = (int)(j / gridDim.x);
            strip = (int)(j - (unsigned)pair * gridDim.x);
        }
    }
#endif
    const int x0 = strip * SW;
    const int xc = clampi(x0 - MH + col, 0, W - 1);
    const size_t plane = (size_t)W * H;
    const Planes R0 = planes_of(R0b + (size_t)pair * pair_stride, plane);
    const Planes R1 = planes_of(R1b + (size_t)pair * pair_stride, plane);
    FlowSrc<UPS> F;
    if constexpr (UPS) {
        F.base = reinterpret_cast<const char*>(flow_in) + (size_t)pair * ups.sw * ups.sh * 8;
        F.sw = ups.sw;
        F.sh = ups.sh;
        F.scale_y = ups.scale_y;
        F.mul = ups.mul;
        lin_x(xc, ups.scale_x, ups.sw, F.sx, F.a1);
        F.a0 = 1.f - F.a1;
        F.c1 = min(F.sx + 1, ups.sw - 1);
    } else {
        F.base = reinterpret_cast<const char*>(flow_in) + (size_t)pair * plane * 8;
        F.W = (unsigned)W;
        F.xc = (unsigned)xc;
    }
    float2* Fout = reinterpret_cast<float2*>(flow_out) + (size_t)pair * plane;
    const int nsteps = (H + 1) / 2;
    // Each role runs its own loop; all three execute one barrier before the loop and two per step.
    const double scale = 1. / (block_size * block_size);
    if constexpr (SPLIT) {
        if (role == 0)
            consumer_loop<MH, COLS, false, true>(mring, sv, R0, R1, F, Fout, fpitch, W, H, x0, xc, col, nsteps, scale);
        else if (role == 1)
            consumer_loop<MH, COLS, true, false>(mring, sv, R0, R1, F, Fout, fpitch, W, H, x0, xc, col, nsteps, scale);
        else if (role == 2)
            producer_loop<MH, COLS, 0, 3>(mring, R0, R1, F, W, H, xc, col, nsteps);
        else
            producer_loop<MH, COLS, 1, 3>(mring, R0, R1, F, W, H, xc, col, nsteps);
    } else {
        if (role == 0)
            consumer_loop<MH, COLS, true, true>(mring, sv, R0, R1, F, Fout, fpitch, W, H, x0, xc, col, nsteps, scale);
        else if (role == 1)
            producer_loop<MH, COLS, 0, 4>(mring, R0, R1, F, W, H, xc, col, nsteps);
        else
            producer_loop<MH, COLS, 1, 4>(mring, R0, R1, F, W, H, xc, col, nsteps);
    }
}


#endif  // NSOF_AB (walker, 2-row producer / consumer kernel)

// ---------------------------------------------------------------------------------------------
// Quad-row variant: the same role-specialised walker, FOUR rows per step.
//
// A step of the 2-row kernel above lasts about as long whatever it carries (measured: ~1.5 us for 128- and for
// 256-column strips): every wave's work per step is one dependent chain -- LDS round trips, the double-precision
// solve, a gather's address arithmetic -- and a CU holds only the 12 waves of one strip (the 100 KB ring), so VALU,
// LDS and the vector L1 all sit at 35-55 % busy while the waves wait on their own previous instruction.  More
// independent work per chain is what fills them: here a step moves 4 rows.
//   waves 0-3   consumers: column sums of 4 rows (thread <-> column), then row sums + solve, thread <-> 4 adjacent
//               pixels of one row (first pixel summed directly, the next three sliding)
//   waves 4-7 / 8-11  producers A / B: rows 0,1 / 2,3 of every step, the loads of their next two steps in flight
// The column sums are single-buffered (4 rows x 5 planes x 256 doubles = 40 KB) so that the ring of 2m+9 rows of M
// (23 x 5 KB at winsize 15) still fits the 160 KB of LDS: two barriers per step (column sums visible / consumed),
// i.e. as many per row as before.  Producers write the first row of step t+1 while the consumers form the column
// sums of step t, and its second row while they solve; the slots those rows overwrite left the window long before.
// Arithmetic per pixel and its order are those of k_iterate_pc (pixels 4k+2, 4k+3 of a row reach their row sums by
// sliding instead of by a direct sum: same double-precision values up to their last bit).
// ---------------------------------------------------------------------------------------------
#ifdef NSOF_Q_TIMING
// Tuning build only (scripts/build_variant.sh ... -DNSOF_Q_TIMING): shader-clock time that one wave of each role of
// workgroup (0,0,0) spends working and waiting at the two barriers of a step; read back by scripts/q_timing.py.
__device__ unsigned long long g_qt[16];
#define QT_ON (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && (threadIdx.x & 255) == 0)
#define QT_DECL unsigned long long qt_prev = __builtin_amdgcn_s_memtime()
#define QT_MARK(slot)                                                      \
    do {                                                                   \
        const unsigned long long qt_now = __builtin_amdgcn_s_memtime();    \
        if (QT_ON) atomicAdd(&g_qt[slot], qt_now - qt_prev);               \
        qt_prev = qt_now;                                                  \
    } while (0)
extern "C" int nsof_debug_qtiming(unsigned long long* out16, int reset)
{
    if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_qt), sizeof(g_qt)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_qt), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#else
#define QT_DECL
#define QT_MARK(slot)
#endif

template <int MH, int COLS_ = 256>
struct QGeom {
    static constexpr int COLS = COLS_, RB = 4;
    static constexpr int RL = 2 * MH + 1 + 2 * RB;
    static constexpr int SW = (COLS - 2 * MH) & ~3;         // a solve thread owns 4 whole pixels
    // Column sums of one (row, plane): the solve threads read 4 adjacent columns each, i.e. lanes 4 doubles apart -- a
    // 4-way bank conflict on a plain row (PMC: 32 % of the kernel's LDS cycles).  Stored as four sub-rows by
    // (column mod 4), 68 doubles apart: a lane's reads (consecutive lanes, consecutive doubles) and the column-sum
    // writes (ds_write_b64: 16 lanes per LDS cycle over 32 banks; 136 dwords = 8 mod 32) are both conflict free.
    static constexpr int SVSUB = COLS / 4 + 4, SVW = 3 * SVSUB + COLS / 4;
    static constexpr size_t SV_BYTES = sizeof(double) * RB * 5 * SVW;
    static constexpr size_t SMEM = SV_BYTES + sizeof(float) * RL * 5 * COLS;
    __host__ __device__ static constexpr int svi(int col) { return (col & 3) * SVSUB + (col >> 2); }
};

template <int MH, int COLS, int GP, int TS, int RR>
__device__ __forceinline__ void q_produce(RowIn (&in)[2][2], FlowSrc<false>::Raw (&fl)[2][2], float (*mring)[5][COLS],
                                          const Planes& R0, const Planes& R1, const FlowSrc<false>& F, int W, int H, int xc,
                                          int col, int t, int yb)
{
    constexpr int RL = QGeom<MH, COLS>::RL;
    const int i = 4 * t + MH + 2 * GP + RR;                  // stream index of this row (row yb + i of the image)
    float Mn[5];
    matrix_from(in[TS][RR], xc, min(yb + i, H - 1), W, H, Mn);
    const int slot = (i + MH + 1) % RL;
#pragma unroll
    for (int c = 0; c < 5; c++) mring[slot][c][col] = Mn[c];
    issue_row(in[TS][RR], R0, R1, W, H, xc, min(yb + i + 8, H - 1), F.resolve(fl[TS][RR]));   // the same row of step t+2
    fl[TS][RR] = F.fetch(min(yb + i + 16, H - 1));                                            // its flow for step t+4
}

template <int MH, int COLS, int GP>
__device__ __forceinline__ void q_producer_loop(float (*mring)[5][COLS], const Planes& R0, const Planes& R1,
                                                const FlowSrc<false>& F, int W, int H, int xc, int col, int nsteps,
                                                int yb)
{
    RowIn in[2][2];
    FlowSrc<false>::Raw fl[2][2];
#pragma unroll
    for (int ts = 0; ts < 2; ts++)
#pragma unroll
        for (int rr = 0; rr < 2; rr++) {
            const int r = min(yb + 4 * ts + MH + 2 * GP + rr, H - 1);
            issue_row(in[ts][rr], R0, R1, W, H, xc, r, F.at(r));
        }
#pragma unroll
    for (int ts = 0; ts < 2; ts++)
#pragma unroll
        for (int rr = 0; rr < 2; rr++) fl[ts][rr] = F.fetch(min(yb + 4 * (ts + 2) + MH + 2 * GP + rr, H - 1));
    // Barriers (all roles alike): B_init, then B1(t), B2(t) for every step t.
    //   before B_init          both rows of step 0
    //   B_init .. B1(0)        first row of step 1            (consumers: column sums of step 0)
    //   B1(t) .. B2(t)         second row of step t+1         (consumers: row sums + solve of step t)
    //   B2(t) .. B1(t+1)       first row of step t+2          (consumers: column sums of step t+1)
    q_produce<MH, COLS, GP, 0, 0>(in, fl, mring, R0, R1, F, W, H, xc, col, 0, yb);
    q_produce<MH, COLS, GP, 0, 1>(in, fl, mring, R0, R1, F, W, H, xc, col, 0, yb);
    __syncthreads();
    q_produce<MH, COLS, GP, 1, 0>(in, fl, mring, R0, R1, F, W, H, xc, col, 1, yb);
    QT_DECL;
    for (int tb = 0; tb < nsteps; tb += 2) {
        QT_MARK(4 + 4 * GP + 2);                                                     // work before B1
        __syncthreads();                                                             // B1(tb)
        QT_MARK(4 + 4 * GP + 3);                                                     // wait at B1
        q_produce<MH, COLS, GP, 1, 1>(in, fl, mring, R0, R1, F, W, H, xc, col, tb + 1, yb);
        QT_MARK(4 + 4 * GP + 0);                                                     // work before B2
        __syncthreads();                                                             // B2(tb)
        QT_MARK(4 + 4 * GP + 1);                                                     // wait at B2
        q_produce<MH, COLS, GP, 0, 0>(in, fl, mring, R0, R1, F, W, H, xc, col, tb + 2, yb);
        if (tb + 1 >= nsteps) break;
        QT_MARK(4 + 4 * GP + 2);
        __syncthreads();                                                             // B1(tb+1)
        QT_MARK(4 + 4 * GP + 3);
        q_produce<MH, COLS, GP, 0, 1>(in, fl, mring, R0, R1, F, W, H, xc, col, tb + 2, yb);
        QT_MARK(4 + 4 * GP + 0);
        __syncthreads();                                                             // B2(tb+1)
        QT_MARK(4 + 4 * GP + 1);
        q_produce<MH, COLS, GP, 1, 0>(in, fl, mring, R0, R1, F, W, H, xc, col, tb + 3, yb);
    }
}

// VOUT: instead of forming the row sums and solving, the column sums of the strip's own columns go to HBM
// (Vout: [H][5][W] doubles of this pair) -- phase A of the exact-order path, see k_rowscan_solve below.
template <int MH, int COLS, bool VOUT = false>
__device__ __forceinline__ void q_consumer_loop(float (*mring)[5][COLS], void* sv_raw, const Planes& R0,
                                                const Planes& R1, const FlowSrc<false>& F, float2* Fout, size_t fpitch,
                                                int W, int H, int x0, int xc, int col, int nsteps, double scale,
                                                double* Vout, int yb, int ye)
{
    using G = QGeom<MH, COLS>;
    constexpr int RL = G::RL, SW = G::SW, TPR = COLS / 4;   // TPR solve threads per row
    double (*sv)[5][G::SVW] = reinterpret_cast<double (*)[5][G::SVW]>(sv_raw);   // [4 rows][5 planes]
    double vs[5];
    if (yb > 0) {
        // a row band that starts inside the image (opt-in NSOF_OPT_ROW_BANDS): the window of row yb-1, rows
        // yb-m-1 .. yb+m-1, is summed directly -- the library's column sums are ONE running sum from row 0, so this
        // start differs from it in the sums' last bits (same class as the row-sum order).
        RowIn t[2];
        const int r0 = max(yb - MH - 1, 0);
        issue_row(t[0], R0, R1, W, H, xc, r0, F.at(r0));
#pragma unroll
        for (int c = 0; c < 5; c++) vs[c] = 0.;
#pragma unroll
        for (int j = 0; j <= 2 * MH; j++) {
            const int r = clampi(yb - MH - 1 + j, 0, H - 1);
            if (j < 2 * MH) {
                const int rn = clampi(yb - MH + j, 0, H - 1);
                issue_row(t[(j + 1) & 1], R0, R1, W, H, xc, rn, F.at(rn));
            }
            float Mi[5];
            matrix_from(t[j & 1], xc, r, W, H, Mi);
#pragma unroll
            for (int c = 0; c < 5; c++) {
                vs[c] += (double)Mi[c];
                mring[j][c][col] = Mi[c];
            }
        }
    } else {
        // prologue: rows 0..m-1 enter the sums; the m+1 rows above the image replicate row 0.
        // ring slot of stream index i is (i + m + 1) % RL.
        RowIn t;
        float M0[5];
        issue_row(t, R0, R1, W, H, xc, 0, F.at(0));
        matrix_from(t, xc, 0, W, H, M0);
#pragma unroll
        for (int c = 0; c < 5; c++) {
            vs[c] = (double)(M0[c] * (float)(MH + 2));   // float product, as "srow0[x]*(m+2)"
#pragma unroll
            for (int j = 0; j <= MH + 1; j++) mring[j][c][col] = M0[c];   // stream indices -m-1 .. 0
        }
#pragma unroll
        for (int i = 1; i < MH; i++) {
            float Mi[5];
            const int r = min(i, H - 1);
            issue_row(t, R0, R1, W, H, xc, r, F.at(r));
            matrix_from(t, xc, r, W, H, Mi);
#pragma unroll
            for (int c = 0; c < 5; c++) {
                vs[c] += (double)Mi[c];
                mring[i + MH + 1][c][col] = Mi[c];
            }
        }
    }
    __syncthreads();   // B_init: step 0 is in the ring
    const int hrow = col / TPR, t4 = col % TPR;   // solve phase: COLS/4 threads per row, 4 pixels each
    int slot_new = (2 * MH + 1) % RL;           // stream index m    -> slot 2m+1
    int slot_old = 0;                           // stream index -m-1 -> slot 0
    const bool own = col >= MH && col < MH + SW && x0 + col - MH < W;   // VOUT: this thread's column belongs to the strip
    QT_DECL;
    for (int t = 0; t < nsteps; t++) {
        // column sums: four more rows enter the window of this thread's column
#pragma unroll
        for (int q = 0; q < 4; q++) {
#pragma unroll
            for (int c = 0; c < 5; c++) {
                const float d = mring[slot_new][c][col] - mring[slot_old][c][col];
                vs[c] += (double)d;
                if constexpr (VOUT) {
                    if (own && yb + 4 * t + q < ye)
                        __builtin_nontemporal_store(vs[c], Vout + ((size_t)(yb + 4 * t + q) * 5 + c) * W + (x0 + col - MH));
                } else {
                    sv[q][c][G::svi(col)] = vs[c];
                }
            }
            slot_new = slot_new + 1 == RL ? 0 : slot_new + 1;
            slot_old = slot_old + 1 == RL ? 0 : slot_old + 1;
        }
        QT_MARK(0);        // column sums
        __syncthreads();   // B1(t): column sums of step t visible
        QT_MARK(1);        // wait at B1
        if constexpr (VOUT) {
            __syncthreads();   // B2(t): same barrier sequence as the solving variant
            continue;
        }
        const int yo = yb + 4 * t + hrow, xo = x0 + 4 * t4;
        if (4 * t4 < SW && yo < ye && xo < W) {
            const double (*svr)[G::SVW] = sv[hrow];
            auto at = [&](int c, int j) { return svr[c][(j & 3) * G::SVSUB + t4 + (j >> 2)]; };   // column 4 t4 + j
            double g[5];
            float2 o[4];
#pragma unroll
            for (int p = 0; p < 4; p++) {
                if (p == 0) {
#pragma unroll
                    for (int c = 0; c < 5; c++) {
                        double a = 0;
#pragma unroll
                        for (int j = 0; j <= 2 * MH; j++) a += at(c, j);
                        g[c] = a;
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < 5; c++) g[c] += at(c, p + 2 * MH) - at(c, p - 1);
                }
                const double g11 = g[0] * scale, g12 = g[1] * scale, g22 = g[2] * scale;
                const double h1 = g[3] * scale, h2 = g[4] * scale;
                const double idet = nsof_recip_normal(g11 * g22 - g12 * g12 + 1e-3);
                o[p].x = (float)((g11 * h2 - g12 * h1) * idet);
                o[p].y = (float)((g22 * h1 - g12 * h2) * idet);
            }
            float2* dst = Fout + (size_t)yo * fpitch + xo;
            if (xo + 3 < W && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
                nsof_store_stream4(reinterpret_cast<float*>(dst), o[0].x, o[0].y, o[1].x, o[1].y);
                nsof_store_stream4(reinterpret_cast<float*>(dst + 2), o[2].x, o[2].y, o[3].x, o[3].y);
            } else {
#pragma unroll
                for (int p = 0; p < 4; p++)
                    if (xo + p < W) dst[p] = o[p];
            }
        }
        QT_MARK(2);        // row sums + solve
        __syncthreads();   // B2(t): column sums consumed, the buffer may be rewritten
        QT_MARK(3);        // wait at B2
    }
}

template <int MH, bool HET, int COLS = 256, bool VOUT = false>
__global__ __launch_bounds__(3 * COLS) void k_iterate_q(const float* __restrict__ R0b, const float* __restrict__ R1b,
                                                    size_t pair_stride, const float* __restrict__ flow_in,
                                                    float* __restrict__ flow_out, int W, int H, int block_size,
                                                    const nsof_het_item* __restrict__ items, int het_final,
                                                    double* __restrict__ vsum_out = nullptr, int band_rows = 0)
{
    using G = QGeom<MH, COLS>;
    constexpr int SW = G::SW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_q[];
    void* sv = smem_q;                                                                          // [4 rows][5][SVW] doubles
    float (*mring)[5][COLS] = reinterpret_cast<float (*)[5][COLS]>(smem_q + G::SV_BYTES);       // [RL]
    const int tid = threadIdx.x, col = tid % COLS;
    const int role = __builtin_amdgcn_readfirstlane(tid / COLS);   // wave-uniform: 0 consumers, 1/2 producers A/B
    int strip = blockIdx.x, pair = blockIdx.z;
    size_t fpitch = (size_t)W;
    if constexpr (HET) {
        const nsof_het_item& it = items[blockIdx.z];
        W = it.wk;
        H = it.hk;
        if (blockIdx.x * SW >= W) return;   // block-uniform, before any barrier
        R0b += it.offR;
        R1b = R0b + 5 * (size_t)W * H;
        pair_stride = 0;
        pair = 0;
        flow_in += 2 * it.offF;
        if constexpr (VOUT) vsum_out += it.offR / 2;
        if (het_final) {
            flow_out = it.out;
            fpitch = (size_t)it.out_pitch;
        } else {
            flow_out += 2 * it.offF;
            fpitch = (size_t)W;
        }
    } else {
#ifndef NSOF_NO_XCD_REMAP
        const unsigned total = gridDim.x * gridDim.z;   // an XCD owns whole pairs (see k_iterate_pc)
        if ((total & 7u) == 0 && gridDim.y == 1) {
            const unsigned lin = blockIdx.x + gridDim.x * blockIdx.z;
            const unsigned j = (lin & 7u) * (total >> 3) + (lin >> 3);
            pair = (int)(j / gridDim.x);
            strip = (int)(j - (unsigned)pair * gridDim.x);
        }
#endif
    }
    const int x0 = strip * SW;
    const int xc = clampi(x0 - MH + col, 0, W - 1);
    const size_t plane = (size_t)W * H;
    const Planes R0 = planes_of(R0b + (size_t)pair * pair_stride, plane);
    const Planes R1 = planes_of(R1b + (size_t)pair * pair_stride, plane);
    FlowSrc<false> F;
    F.base = reinterpret_cast<const char*>(flow_in) + (size_t)pair * plane * 8;
    F.W = (unsigned)W;
    F.xc = (unsigned)xc;
    float2* Fout = reinterpret_cast<float2*>(flow_out) + (size_t)pair * plane;
    // band_rows > 0 (a multiple of 4): blockIdx.y owns rows [yb, ye) only -- more workgroups for a small batch
    int yb = 0, ye = H;
    if (band_rows > 0) {
        yb = blockIdx.y * band_rows;
        ye = min(H, yb + band_rows);
        if (yb >= H) return;   // block-uniform, before any barrier
    }
    const int nsteps = (ye - yb + 3) / 4;
    if (role == 0)
        q_consumer_loop<MH, COLS, VOUT>(mring, sv, R0, R1, F, Fout, fpitch, W, H, x0, xc, col, nsteps,
                                        1. / (block_size * block_size), VOUT ? vsum_out + (size_t)pair * 5 * plane : nullptr,
                                        yb, ye);
    else if (role == 1)
        q_producer_loop<MH, COLS, 0>(mring, R0, R1, F, W, H, xc, col, nsteps, yb);
    else
        q_producer_loop<MH, COLS, 1>(mring, R0, R1, F, W, H, xc, col, nsteps, yb);
}

#ifdef NSOF_AB   // tuning builds only: round 2's two-kernel form of the library's row-sum order (NSOF_EXACT_IMPL=2k, NSOF_LAT_ROWSCAN_OLD);
                  // the product library keeps that order inside k_iterate_x / the k_lat_* kernels
// ---------------------------------------------------------------------------------------------
// Exact-order row sums, phase B: the reference library's running row sums + the 2x2 solve.
//
// The library forms the box filter's ROW sums as ONE running double-precision sum along each image row,
// g += vsum[x+m] - vsum[x-m-1]; where the 2x2 system is rank deficient the rounding history of that sum decides the
// flow's 4th decimal (and much more on degenerate frames), so bit-level parity needs this very order
// (DESIGN.md section 2).  It is sequential along x over the whole row, hence a second kernel: phase A
// (k_iterate_q<.., VOUT>) leaves the column sums in HBM, [H][5][W] doubles per pair; here a workgroup owns a band of
// RS_ROWS rows and walks the image left to right in tiles of 8 columns:
//   load   the tile's column sums -> an LDS ring of 32 columns (64-B row segments)
//   scan   thread <-> (row, plane): 8 steps of the running sum, kept in a register, results -> LDS
//   solve  thread <-> pixel of the 64 x 8 tile: the 2x2 solve, flow stored as 64-B row segments
// The next tile's loads are in flight during the solve.  Same arithmetic and order as the library's FarnebackUpdateFlow_Blur, bit for bit.
// ---------------------------------------------------------------------------------------------
#ifndef NSOF_RS_ROWS
#define NSOF_RS_ROWS 32
#endif
constexpr int RS_ROWS = NSOF_RS_ROWS, RS_TW = 8, RS_RING = 32, RS_VSTR = 33, RS_SSTR = 9;
constexpr int RS_THREADS = 5 * RS_ROWS;                      // thread <-> (plane, row) in the scan
constexpr int RS_RSH = RS_ROWS == 64 ? 6 : 5;                // log2(RS_ROWS)
static_assert(RS_ROWS == 32 || RS_ROWS == 64, "row band of 32 or 64 rows");
constexpr size_t RS_SMEM = sizeof(double) * 5 * RS_ROWS * (RS_VSTR + RS_SSTR);

__global__ __launch_bounds__(RS_THREADS) void k_rowscan_solve(const double* __restrict__ V, int W, int H, int m, int block_size,
                                                       float* __restrict__ flow, size_t fpitch_default,
                                                       const nsof_het_item* __restrict__ items, int het_final)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_rs[];
    double (*Vw)[RS_ROWS][RS_VSTR] = reinterpret_cast<double (*)[RS_ROWS][RS_VSTR]>(smem_rs);                  // [5]
    double (*St)[RS_ROWS][RS_SSTR] =
        reinterpret_cast<double (*)[RS_ROWS][RS_SSTR]>(smem_rs + sizeof(double) * 5 * RS_ROWS * RS_VSTR);      // [5]
    const int tid = threadIdx.x;
    size_t fpitch = fpitch_default;
    float2* Fout;
    const double* Vp;
    if (items) {   // work list: blockIdx.z indexes the level's item table
        const nsof_het_item& it = items[blockIdx.z];
        W = it.wk;
        H = it.hk;
        if (blockIdx.x * RS_ROWS >= H) return;
        Vp = V + it.offR / 2;   // an item's column sums (5 wk hk doubles) mirror its expansion block (10 wk hk floats)
        if (het_final) {
            Fout = reinterpret_cast<float2*>(it.out);
            fpitch = (size_t)it.out_pitch;
        } else {
            Fout = reinterpret_cast<float2*>(flow) + it.offF;
            fpitch = (size_t)W;
        }
    } else {
        Vp = V + (size_t)blockIdx.z * 5 * W * H;
        Fout = reinterpret_cast<float2*>(flow) + (size_t)blockIdx.z * W * H;
    }
    const int y0 = blockIdx.x * RS_ROWS;
    const double scale = 1. / (block_size * block_size);
    // loader: element e = tid + RS_THREADS i of an 8-column tile: column e % 8, row (e / 8) % RS_ROWS, plane e / (8 RS_ROWS)
    auto load_cols = [&](int xa, double (&reg)[8]) {   // columns [xa, xa + 8) -> registers
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int e = tid + RS_THREADS * i;
            const int cx = e & 7, r = (e >> 3) & (RS_ROWS - 1), c = e >> (3 + RS_RSH);
            const int y = min(y0 + r, H - 1), x = min(xa + cx, W - 1);
            reg[i] = Vp[((size_t)y * 5 + c) * W + x];
        }
    };
    auto store_cols = [&](int xa, const double (&reg)[8]) {   // columns beyond the image keep their slots' contents
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int e = tid + RS_THREADS * i;
            const int cx = e & 7, r = (e >> 3) & (RS_ROWS - 1), c = e >> (3 + RS_RSH);
            if (xa + cx < W) Vw[c][r][(xa + cx) & (RS_RING - 1)] = reg[i];
        }
    };
    // The scan of the tile at xt reads columns [xt - m - 1, xt + 7 + m] (clamped to the image); the ring holds
    // [loaded - 32, loaded): loaded starts at 24 and grows by 8 per tile, so xt + 8 + m <= loaded <= xt + 32 (m <= 7).
    // The ring has room for ONE tile of lookahead only, so the HBM latency is covered in registers: three sets of 8
    // values per thread are in flight, a set is written to the ring three tiles after its loads were issued.
    double regs[3][8];
    for (int k = 0; k < 3; k++) {
        load_cols(RS_TW * k, regs[0]);
        store_cols(RS_TW * k, regs[0]);
    }
#pragma unroll
    for (int k = 0; k < 3; k++) load_cols(24 + RS_TW * k, regs[k]);   // columns of tiles +3, +4, +5 on their way
    __syncthreads();
    const int sc = tid >> RS_RSH, sr = tid & (RS_ROWS - 1);   // scan role: (plane, row)
    auto col = [&](int x) { return Vw[sc][sr][min(max(x, 0), W - 1) & (RS_RING - 1)]; };
    double S = col(0) * (m + 2);
    for (int x = 1; x < m; x++) S += col(x);
    auto tile = [&](auto kc, int xt) {
        constexpr int K = decltype(kc)::value;   // register set of this tile's refill
        // scan: the library's running sum over this tile's columns
#pragma unroll
        for (int j = 0; j < RS_TW; j++) {
            const int x = xt + j;
            S += col(x + m) - col(x - m - 1);
            St[sc][sr][j] = S;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int e = tid + RS_THREADS * i;
            if (e < RS_ROWS * RS_TW) {
                const int px = e & 7, r = e >> 3;
                const int x = xt + px, y = y0 + r;
                if (x < W && y < H) {
                    const double g11 = St[0][r][px] * scale, g12 = St[1][r][px] * scale, g22 = St[2][r][px] * scale;
                    const double h1 = St[3][r][px] * scale, h2 = St[4][r][px] * scale;
                    const double idet = nsof_recip_normal(g11 * g22 - g12 * g12 + 1e-3);
                    Fout[(size_t)y * fpitch + x] =
                        make_float2((float)((g11 * h2 - g12 * h1) * idet), (float)((g22 * h1 - g12 * h2) * idet));
                }
            }
        }
        // columns [xt + 24, xt + 32) (issued three tiles ago) take the slots of [xt - 8, xt), which only the scan
        // above still needed; then the set is refilled with the columns three tiles further on
        store_cols(xt + 24, regs[K]);
        load_cols(xt + 48, regs[K]);
        __syncthreads();
    };
    for (int xt = 0; xt < W; xt += 3 * RS_TW) {
        tile(std::integral_constant<int, 0>{}, xt);
        if (xt + RS_TW < W) tile(std::integral_constant<int, 1>{}, xt + RS_TW);
        if (xt + 2 * RS_TW < W) tile(std::integral_constant<int, 2>{}, xt + 2 * RS_TW);
    }
}

template <int MH>
int launch_iterate_q_exact(nsof_ctx* ctx, int n_pairs, const float* R0, const float* R1, size_t pair_stride,
                           const float* flow_in, float* flow_out, int W, int H, int winsize, double* vsum)
{
    using G = QGeom<MH>;
    if (int rc = lds_opt_in(ctx, k_iterate_q<MH, false, 256, true>, G::SMEM)) return rc;
    if (int rc = lds_opt_in(ctx, k_rowscan_solve, RS_SMEM)) return rc;
    dim3 grid((W + G::SW - 1) / G::SW, 1, n_pairs);
    hipLaunchKernelGGL((k_iterate_q<MH, false, 256, true>), grid, dim3(768), G::SMEM, ctx->stream, R0, R1, pair_stride,
                       flow_in, flow_out, W, H, winsize, nullptr, 0, vsum);
    hipLaunchKernelGGL(k_rowscan_solve, dim3((H + RS_ROWS - 1) / RS_ROWS, 1, n_pairs), dim3(RS_THREADS), RS_SMEM, ctx->stream,
                       (const double*)vsum, W, H, MH, winsize, flow_out, (size_t)W, nullptr, 0);
    return NSOF_OK;
}

template <int MH>
int launch_iterate_q_het_exact(nsof_ctx* ctx, int n_items, const nsof_het_item* items, int max_w, int max_h, const float* R,
                               const float* flow_in, float* flow_out, bool final, int winsize, double* vsum)
{
    using G = QGeom<MH>;
    if (int rc = lds_opt_in(ctx, k_iterate_q<MH, true, 256, true>, G::SMEM)) return rc;
    if (int rc = lds_opt_in(ctx, k_rowscan_solve, RS_SMEM)) return rc;
    dim3 grid((max_w + G::SW - 1) / G::SW, 1, n_items);
    hipLaunchKernelGGL((k_iterate_q<MH, true, 256, true>), grid, dim3(768), G::SMEM, ctx->stream, R, R, (size_t)0, flow_in,
                       flow_out, 0, 0, winsize, items, 0, vsum);
    hipLaunchKernelGGL(k_rowscan_solve, dim3((max_h + RS_ROWS - 1) / RS_ROWS, 1, n_items), dim3(RS_THREADS), RS_SMEM, ctx->stream,
                       (const double*)vsum, 0, 0, MH, winsize, flow_out, (size_t)0, items, final ? 1 : 0);
    return NSOF_OK;
}

#endif  // NSOF_AB (two-kernel exact form)

template <int MH>
int launch_iterate_q(nsof_ctx* ctx, int n_pairs, const float* R0, const float* R1, size_t pair_stride,
                     const float* flow_in, float* flow_out, int W, int H, int winsize)
{
#ifdef NSOF_AB
    static const bool narrow = NSOF_AB_GETENV("NSOF_Q_COLS128") != nullptr;   // A/B: two 128-column strips per CU
    if (narrow) {
        using G = QGeom<MH, 128>;
        if (int rc = lds_opt_in(ctx, k_iterate_q<MH, false, 128>, G::SMEM)) return rc;
        dim3 grid((W + G::SW - 1) / G::SW, 1, n_pairs);
        hipLaunchKernelGGL((k_iterate_q<MH, false, 128>), grid, dim3(384), G::SMEM, ctx->stream, R0, R1, pair_stride,
                           flow_in, flow_out, W, H, winsize, nullptr, 0);
        return NSOF_OK;
    }
#endif
    using G = QGeom<MH>;
    if (int rc = lds_opt_in(ctx, k_iterate_q<MH, false>, G::SMEM)) return rc;
    dim3 grid((W + G::SW - 1) / G::SW, 1, n_pairs);
    // Opt-in row bands (NSOF_OPT_ROW_BANDS): a small batch has too few (strip, pair) workgroups for 256 CUs and each
    // walks the whole height; bands of rows add workgroups at the price of 2m+1 extra rows per band.  1 = automatic
    // (bands no shorter than 32 rows, until the launch has about two workgroups per CU), >= 4 = that many rows.
    // Automatic mode only from winsize 9 up: with small windows the 2x2 systems are rank deficient often enough that a
    // band's restart shows in the 4th decimal of many pixels (parity soak, docs/HISTORY_r1_r3.md section 5.1); an explicit row
    // count is taken at its word.
    int band_rows = 0;
    if (ctx->opt_row_bands > 0) {
        if (ctx->opt_row_bands >= 4) {
            band_rows = (ctx->opt_row_bands + 3) & ~3;
        } else if (winsize < 9) {
            band_rows = 0;
        } else {
            const int want = (512 + (int)(grid.x * grid.z) - 1) / (int)(grid.x * grid.z);   // bands per strip
            band_rows = std::max(32, ((H + want - 1) / want + 3) & ~3);
        }
        if (band_rows >= H) band_rows = 0;
    }
    if (band_rows > 0) grid.y = (H + band_rows - 1) / band_rows;
    hipLaunchKernelGGL((k_iterate_q<MH, false>), grid, dim3(768), G::SMEM, ctx->stream, R0, R1, pair_stride, flow_in,
                       flow_out, W, H, winsize, nullptr, 0, nullptr, band_rows);
    return NSOF_OK;
}

template <int MH>
int launch_iterate_q_het(nsof_ctx* ctx, int n_items, const nsof_het_item* items, int max_w, const float* R,
                         const float* flow_in, float* flow_out, bool final, int winsize)
{
    using G = QGeom<MH>;
    if (int rc = lds_opt_in(ctx, k_iterate_q<MH, true>, G::SMEM)) return rc;
    dim3 grid((max_w + G::SW - 1) / G::SW, 1, n_items);
    hipLaunchKernelGGL((k_iterate_q<MH, true>), grid, dim3(768), G::SMEM, ctx->stream, R, R, (size_t)0, flow_in, flow_out,
                       0, 0, winsize, items, final ? 1 : 0);
    return NSOF_OK;
}

#ifdef NSOF_AB
template <int MH, int COLS, bool UPS>
int launch_iterate_pc_c(nsof_ctx* ctx, int n_pairs, const float* R0, const float* R1, size_t pair_stride,
                        const float* flow_in, float* flow_out, int W, int H, int winsize, const UpsArgs& ups)
{
    using G = PCGeom<MH, COLS>;
    dim3 grid((W + G::SW - 1) / G::SW, 1, n_pairs);
    if constexpr (!UPS && COLS == 256) {
        static const bool split = NSOF_AB_GETENV("NSOF_ITER_SPLIT") != nullptr;   // 16-wave layout (A/B)
        if (split) {
            if (int rc = lds_opt_in(ctx, k_iterate_pc<MH, COLS, false, false, true>, G::SMEM)) return rc;
            hipLaunchKernelGGL((k_iterate_pc<MH, COLS, false, false, true>), grid, dim3(4 * COLS), G::SMEM, ctx->stream,
                               R0, R1, pair_stride, flow_in, flow_out, W, H, winsize, ups, nullptr, 0);
            return NSOF_OK;
        }
    }
    if (int rc = lds_opt_in(ctx, k_iterate_pc<MH, COLS, UPS>, G::SMEM)) return rc;
    hipLaunchKernelGGL((k_iterate_pc<MH, COLS, UPS>), grid, dim3(3 * COLS), G::SMEM, ctx->stream, R0, R1, pair_stride,
                       flow_in, flow_out, W, H, winsize, ups);
    return NSOF_OK;
}

template <int MH>
int launch_iterate_pc_het(nsof_ctx* ctx, int n_items, const nsof_het_item* items, int max_w, const float* R,
                          const float* flow_in, float* flow_out, bool final, int winsize)
{
    using G = PCGeom<MH, 256>;
    if (int rc = lds_opt_in(ctx, k_iterate_pc<MH, 256, false, true>, G::SMEM)) return rc;
    dim3 grid((max_w + G::SW - 1) / G::SW, 1, n_items);
    hipLaunchKernelGGL((k_iterate_pc<MH, 256, false, true>), grid, dim3(3 * 256), G::SMEM, ctx->stream, R, R, (size_t)0,
                       flow_in, flow_out, 0, 0, winsize, UpsArgs{}, items, final ? 1 : 0);
    return NSOF_OK;
}

template <int MH, bool UPS>
int launch_iterate_pc(nsof_ctx* ctx, int n_pairs, const float* R0, const float* R1, size_t pair_stride,
                       const float* flow_in, float* flow_out, int W, int H, int winsize, const UpsArgs& ups)
{
    static const bool narrow = NSOF_AB_GETENV("NSOF_PC_COLS128") != nullptr;   // tuning experiment: 2 blocks of 128 columns per CU
    if (!UPS && narrow)
        return launch_iterate_pc_c<MH, 128, false>(ctx, n_pairs, R0, R1, pair_stride, flow_in, flow_out, W, H, winsize, ups);
    return launch_iterate_pc_c<MH, 256, UPS>(ctx, n_pairs, R0, R1, pair_stride, flow_in, flow_out, W, H, winsize, ups);
}

template <bool UPS>
int launch_iterate_pc_m(nsof_ctx* ctx, int m, int n_pairs, const float* R0, const float* R1, size_t pair_stride,
                         const float* flow_in, float* flow_out, int W, int H, int winsize, const UpsArgs& ups)
{
    switch (m) {
        case 1: return launch_iterate_pc<1, UPS>(ctx, n_pairs, R0, R1, pair_stride, flow_in, flow_out, W, H, winsize, ups);
        case 2: return launch_iterate_pc<2, UPS>(ctx, n_pairs, R0, R1, pair_stride, flow_in, flow_out, W, H, winsize, ups);
        case 3: return launch_iterate_pc<3, UPS>(ctx, n_pairs, R0, R1, pair_stride, flow_in, flow_out, W, H, winsize, ups);
        case 4: return launch_iterate_pc<4, UPS>(ctx, n_pairs, R0, R1, pair_stride, flow_in, flow_out, W, H, winsize, ups);
        case 5: return launch_iterate_pc<5, UPS>(ctx, n_pairs, R0, R1, pair_stride, flow_in, flow_out, W, H, winsize, ups);
        case 6: return launch_iterate_pc<6, UPS>(ctx, n_pairs, R0, R1, pair_stride, flow_in, flow_out, W, H, winsize, ups);
        default: return launch_iterate_pc<7, UPS>(ctx, n_pairs, R0, R1, pair_stride, flow_in, flow_out, W, H, winsize, ups);
    }
}

template <int MH>
void launch_iterate_m(nsof_ctx* ctx, int n_pairs, const float* R0, const float* R1, size_t pair_stride,
                      const float* flow_in, float* flow_out, int W, int H, int winsize)
{
    constexpr int SW = IterGeom<MH>::SW;
    dim3 grid((W + SW - 1) / SW, 1, n_pairs);
    hipLaunchKernelGGL(k_iterate<MH>, grid, dim3(256), 0, ctx->stream, R0, R1, pair_stride, flow_in, flow_out, W, H,
                       winsize);
}

#endif  // NSOF_AB

}  // namespace

bool nsof_iterate_supported(int winsize, int W, int H)
{
    const int m = winsize / 2;
#ifdef NSOF_AB
    return m >= 1 && m <= 8 && W >= 2 && H >= 2;   // (the walker of the tuning builds also takes winsize 16 / 17)
#else
    return m >= 1 && m <= 7 && W >= 2 && H >= 2;   // the clamped gather needs a 2x2 neighbourhood to exist
#endif
}

#ifdef NSOF_AB
static int g_iterate_variant = -1;   // NSOF_ITERATE=walker|pc|quad (tuning / A-B runs); default quad
static int iterate_variant()
{
    if (g_iterate_variant < 0) {
        const char* e = NSOF_AB_GETENV("NSOF_ITERATE");
        g_iterate_variant = (e && e[0] == 'w') ? 0 : (e && e[0] == 'p') ? 1 : 2;
    }
    return g_iterate_variant;
}
static bool use_pc(int m) { return iterate_variant() >= 1 && m <= 7; }
static bool use_quad(int m) { return iterate_variant() == 2 && m <= 7; }
#else
static bool use_quad(int m) { return m <= 7; }
#endif

template <typename... A>
static int launch_iterate_q_m(int m, A... a)
{
    switch (m) {
        case 1: return launch_iterate_q<1>(a...);
        case 2: return launch_iterate_q<2>(a...);
        case 3: return launch_iterate_q<3>(a...);
        case 4: return launch_iterate_q<4>(a...);
        case 5: return launch_iterate_q<5>(a...);
        case 6: return launch_iterate_q<6>(a...);
        default: return launch_iterate_q<7>(a...);
    }
}
template <typename... A>
static int launch_iterate_q_het_m(int m, A... a)
{
    switch (m) {
        case 1: return launch_iterate_q_het<1>(a...);
        case 2: return launch_iterate_q_het<2>(a...);
        case 3: return launch_iterate_q_het<3>(a...);
        case 4: return launch_iterate_q_het<4>(a...);
        case 5: return launch_iterate_q_het<5>(a...);
        case 6: return launch_iterate_q_het<6>(a...);
        default: return launch_iterate_q_het<7>(a...);
    }
}

// flow_in and flow_out must be different buffers (rows y+m of flow_in are read while row y of flow_out is written).
int nsof_launch_iterate(nsof_ctx* ctx, int n_pairs, const float* R0, const float* R1, size_t pair_stride,
                        const float* flow_in, float* flow_out, int W, int H, int winsize)
{
    nsof_prof_scope ps(ctx, NSOF_K_ITERATE);
    if (use_quad(winsize / 2)) {
        if (int rc = launch_iterate_q_m(winsize / 2, ctx, n_pairs, R0, R1, pair_stride, flow_in, flow_out, W, H, winsize))
            return rc;
        NSOF_HIP(ctx, hipGetLastError());
        return NSOF_OK;
    }
#ifdef NSOF_AB
    if (use_pc(winsize / 2)) {
        if (int rc = launch_iterate_pc_m<false>(ctx, winsize / 2, n_pairs, R0, R1, pair_stride, flow_in, flow_out, W, H,
                                                winsize, UpsArgs{}))
            return rc;
        NSOF_HIP(ctx, hipGetLastError());
        return NSOF_OK;
    }
    switch (winsize / 2) {
        case 1: launch_iterate_m<1>(ctx, n_pairs, R0, R1, pair_stride, flow_in, flow_out, W, H, winsize); break;
        case 2: launch_iterate_m<2>(ctx, n_pairs, R0, R1, pair_stride, flow_in, flow_out, W, H, winsize); break;
        case 3: launch_iterate_m<3>(ctx, n_pairs, R0, R1, pair_stride, flow_in, flow_out, W, H, winsize); break;
        case 4: launch_iterate_m<4>(ctx, n_pairs, R0, R1, pair_stride, flow_in, flow_out, W, H, winsize); break;
        case 5: launch_iterate_m<5>(ctx, n_pairs, R0, R1, pair_stride, flow_in, flow_out, W, H, winsize); break;
        case 6: launch_iterate_m<6>(ctx, n_pairs, R0, R1, pair_stride, flow_in, flow_out, W, H, winsize); break;
        case 7: launch_iterate_m<7>(ctx, n_pairs, R0, R1, pair_stride, flow_in, flow_out, W, H, winsize); break;
        case 8: launch_iterate_m<8>(ctx, n_pairs, R0, R1, pair_stride, flow_in, flow_out, W, H, winsize); break;
        default: return nsof_set_error(ctx, NSOF_EUNSUPPORTED, "fused iteration supports winsize 2..17");
    }
    NSOF_HIP(ctx, hipGetLastError());
    return NSOF_OK;
#else
    return nsof_set_error(ctx, NSOF_EUNSUPPORTED, "fused iteration supports winsize 2..15");
#endif
}

// First iteration of a level, reading the previous (coarser) level's flow and resampling it on the fly (tuning builds
// only: NSOF_FOLD_UPSAMPLE measured slower than the standalone resample kernel, docs/HISTORY_r1_r3.md section 5).
bool nsof_iterate_upsample_supported(int winsize, int W, int H)
{
#ifdef NSOF_AB
    return nsof_iterate_supported(winsize, W, H) && use_pc(winsize / 2);
#else
    (void)winsize; (void)W; (void)H;
    return false;
#endif
}

int nsof_launch_iterate_upsample(nsof_ctx* ctx, int n_pairs, const float* R0, const float* R1, size_t pair_stride,
                                 const float* coarse_flow, int sw, int sh, float mul, float* flow_out, int W, int H,
                                 int winsize)
{
    if (!nsof_iterate_upsample_supported(winsize, W, H))
        return nsof_set_error(ctx, NSOF_EUNSUPPORTED, "fused upsample+iteration not available for winsize %d", winsize);
#ifdef NSOF_AB
    nsof_prof_scope ps(ctx, NSOF_K_ITERATE);
    UpsArgs ups;
    ups.sw = sw;
    ups.sh = sh;
    ups.scale_x = 1. / ((double)W / sw);
    ups.scale_y = 1. / ((double)H / sh);
    ups.mul = mul;
    if (int rc = launch_iterate_pc_m<true>(ctx, winsize / 2, n_pairs, R0, R1, pair_stride, coarse_flow, flow_out, W, H,
                                           winsize, ups))
        return rc;
    NSOF_HIP(ctx, hipGetLastError());
#endif
    return NSOF_OK;
}

// Work-list twin of nsof_launch_iterate (role-specialised kernel only: winsize 2..15).
int nsof_launch_iterate_het(nsof_ctx* ctx, int n_items, const nsof_het_item* d_items, int max_w, const float* R,
                            const float* flow_in, float* flow_out, bool final, int winsize)
{
    const int m = winsize / 2;
    if (m < 1 || m > 7) return nsof_set_error(ctx, NSOF_EUNSUPPORTED, "work-list iteration supports winsize 2..15");
    nsof_prof_scope ps(ctx, NSOF_K_ITERATE);
    int rc;
    if (use_quad(m)) {
        if ((rc = launch_iterate_q_het_m(m, ctx, n_items, d_items, max_w, R, flow_in, flow_out, final, winsize))) return rc;
        NSOF_HIP(ctx, hipGetLastError());
        return NSOF_OK;
    }
#ifndef NSOF_AB
    return nsof_set_error(ctx, NSOF_EUNSUPPORTED, "work-list iteration supports winsize 2..15");
#else
    switch (m) {
        case 1: rc = launch_iterate_pc_het<1>(ctx, n_items, d_items, max_w, R, flow_in, flow_out, final, winsize); break;
        case 2: rc = launch_iterate_pc_het<2>(ctx, n_items, d_items, max_w, R, flow_in, flow_out, final, winsize); break;
        case 3: rc = launch_iterate_pc_het<3>(ctx, n_items, d_items, max_w, R, flow_in, flow_out, final, winsize); break;
        case 4: rc = launch_iterate_pc_het<4>(ctx, n_items, d_items, max_w, R, flow_in, flow_out, final, winsize); break;
        case 5: rc = launch_iterate_pc_het<5>(ctx, n_items, d_items, max_w, R, flow_in, flow_out, final, winsize); break;
        case 6: rc = launch_iterate_pc_het<6>(ctx, n_items, d_items, max_w, R, flow_in, flow_out, final, winsize); break;
        default: rc = launch_iterate_pc_het<7>(ctx, n_items, d_items, max_w, R, flow_in, flow_out, final, winsize); break;
    }
    if (rc) return rc;
    NSOF_HIP(ctx, hipGetLastError());
    return NSOF_OK;
#endif
}

// Round 2's two-kernel form of the library's row-sum order (tuning builds only; see the NSOF_AB section above).
#ifdef NSOF_AB
// Exact-order twin of nsof_launch_iterate: phase A (fused matrix update + column sums -> vsum) and phase B (row scan
// + solve).  vsum: n_pairs * 5 * W * H doubles of scratch.  winsize 2..15.
// Phase B on its own (the small-batch form, farneback_iterate_lat.hip, forms the column sums with its own kernels).
int nsof_launch_rowscan_solve(nsof_ctx* ctx, int n_pairs, const double* V, int W, int H, int winsize, float* flow_out)
{
    nsof_prof_scope ps(ctx, NSOF_K_ITERATE);
    if (int rc = lds_opt_in(ctx, k_rowscan_solve, RS_SMEM)) return rc;
    hipLaunchKernelGGL(k_rowscan_solve, dim3((H + RS_ROWS - 1) / RS_ROWS, 1, n_pairs), dim3(RS_THREADS), RS_SMEM, ctx->stream, V,
                       W, H, winsize / 2, winsize, flow_out, (size_t)W, nullptr, 0);
    NSOF_HIP(ctx, hipGetLastError());
    return NSOF_OK;
}

int nsof_launch_rowscan_solve_het(nsof_ctx* ctx, int n_items, const nsof_het_item* items, int max_h, const double* V,
                                  float* flow_out, bool final, int winsize)
{
    nsof_prof_scope ps(ctx, NSOF_K_ITERATE);
    if (int rc = lds_opt_in(ctx, k_rowscan_solve, RS_SMEM)) return rc;
    hipLaunchKernelGGL(k_rowscan_solve, dim3((max_h + RS_ROWS - 1) / RS_ROWS, 1, n_items), dim3(RS_THREADS), RS_SMEM, ctx->stream,
                       V, 0, 0, winsize / 2, winsize, flow_out, (size_t)0, items, final ? 1 : 0);
    NSOF_HIP(ctx, hipGetLastError());
    return NSOF_OK;
}

bool nsof_iterate_exact_supported(int winsize, int W, int H)
{
    const int m = winsize / 2;
    return m >= 1 && m <= 7 && W >= 2 && H >= 2;
}

int nsof_launch_iterate_exact(nsof_ctx* ctx, int n_pairs, const float* R0, const float* R1, size_t pair_stride,
                              const float* flow_in, float* flow_out, int W, int H, int winsize, double* vsum)
{
    nsof_prof_scope ps(ctx, NSOF_K_ITERATE);
    int rc;
    switch (winsize / 2) {
        case 1: rc = launch_iterate_q_exact<1>(ctx, n_pairs, R0, R1, pair_stride, flow_in, flow_out, W, H, winsize, vsum); break;
        case 2: rc = launch_iterate_q_exact<2>(ctx, n_pairs, R0, R1, pair_stride, flow_in, flow_out, W, H, winsize, vsum); break;
        case 3: rc = launch_iterate_q_exact<3>(ctx, n_pairs, R0, R1, pair_stride, flow_in, flow_out, W, H, winsize, vsum); break;
        case 4: rc = launch_iterate_q_exact<4>(ctx, n_pairs, R0, R1, pair_stride, flow_in, flow_out, W, H, winsize, vsum); break;
        case 5: rc = launch_iterate_q_exact<5>(ctx, n_pairs, R0, R1, pair_stride, flow_in, flow_out, W, H, winsize, vsum); break;
        case 6: rc = launch_iterate_q_exact<6>(ctx, n_pairs, R0, R1, pair_stride, flow_in, flow_out, W, H, winsize, vsum); break;
        case 7: rc = launch_iterate_q_exact<7>(ctx, n_pairs, R0, R1, pair_stride, flow_in, flow_out, W, H, winsize, vsum); break;
        default: return nsof_set_error(ctx, NSOF_EUNSUPPORTED, "exact-order fused iteration supports winsize 2..15");
    }
    if (rc) return rc;
    NSOF_HIP(ctx, hipGetLastError());
    return NSOF_OK;
}

// Work-list twin of nsof_launch_iterate_exact; vsum mirrors the level's expansion buffer (same size in bytes).
int nsof_launch_iterate_het_exact(nsof_ctx* ctx, int n_items, const nsof_het_item* d_items, int max_w, int max_h,
                                  const float* R, const float* flow_in, float* flow_out, bool final, int winsize,
                                  double* vsum)
{
    nsof_prof_scope ps(ctx, NSOF_K_ITERATE);
    int rc;
    switch (winsize / 2) {
#define NSOF_QHE(MM) case MM: rc = launch_iterate_q_het_exact<MM>(ctx, n_items, d_items, max_w, max_h, R, flow_in, flow_out, final, winsize, vsum); break
        NSOF_QHE(1); NSOF_QHE(2); NSOF_QHE(3); NSOF_QHE(4); NSOF_QHE(5); NSOF_QHE(6); NSOF_QHE(7);
#undef NSOF_QHE
        default: return nsof_set_error(ctx, NSOF_EUNSUPPORTED, "exact-order work-list iteration supports winsize 2..15");
    }
    if (rc) return rc;
    NSOF_HIP(ctx, hipGetLastError());
    return NSOF_OK;
}

#else
bool nsof_iterate_exact_supported(int, int, int) { return false; }
int nsof_launch_rowscan_solve(nsof_ctx* ctx, int, const double*, int, int, int, float*)
{
    return nsof_set_error(ctx, NSOF_EUNSUPPORTED, "two-kernel exact form: tuning builds only");
}
int nsof_launch_rowscan_solve_het(nsof_ctx* ctx, int, const nsof_het_item*, int, const double*, float*, bool, int)
{
    return nsof_set_error(ctx, NSOF_EUNSUPPORTED, "two-kernel exact form: tuning builds only");
}
int nsof_launch_iterate_exact(nsof_ctx* ctx, int, const float*, const float*, size_t, const float*, float*, int, int, int, double*)
{
    return nsof_set_error(ctx, NSOF_EUNSUPPORTED, "two-kernel exact form: tuning builds only");
}
int nsof_launch_iterate_het_exact(nsof_ctx* ctx, int, const nsof_het_item*, int, int, const float*, const float*, float*, bool, int,
                                  double*)
{
    return nsof_set_error(ctx, NSOF_EUNSUPPORTED, "two-kernel exact form: tuning builds only");
}
#endif
