// Internal declarations shared by the HIP translation units of libnsof.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <vector>

#include "nsof.h"

#define NSOF_MAX_POLY_N 10    // templated fast kernels exist for radius 1..10 (reference uses 1, 5, 10)
#define NSOF_MAX_BLUR_TAPS 64 // pyramid Gaussian kernel size limit (reference needs <= 19)

struct nsof_prof_slot {
    std::vector<hipEvent_t> start, stop;  // event pool, reused
    size_t used = 0;
    double acc_ms = 0;       // already-collected time
    long long acc_launches = 0;
};

struct nsof_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t own_stream = nullptr;
    unsigned prof_mask = 0;
    nsof_prof_slot prof[NSOF_K_COUNT];
    int opt_polyexp_f32 = 0;   // NSOF_OPT_POLYEXP_F32
    int opt_exact_rowsums = 1; // NSOF_OPT_EXACT_ROWSUMS (default: the library's row-sum order)
    int opt_row_bands = 0;     // NSOF_OPT_ROW_BANDS: 0 off, 1 automatic, >= 4 rows per band
    int opt_small_batch_jobs = 64;    // NSOF_OPT_SMALL_BATCH_JOBS: calls with at most this many (strip, image) jobs take the three-kernel exact form
    int dbg_fault = 0;         // NSOF_OPT_DEBUG_FAULT (test hook): bit 0 k_iterate_x withholds a carry, bit 1 k_lat_colsum a turn
    int opt_pyr_fma = 0;       // NSOF_OPT_PYR_FMA: pyramid blur / resamples with fused multiply-adds (arithmetic variant twin)
    char err[512] = {0};
    // reusable device workspace of the Farneback driver
    void* ws = nullptr;
    size_t ws_bytes = 0;
    // staging for the host-pointer entry point
    void* stage = nullptr;
    size_t stage_bytes = 0;
    // pinned host staging of the host-pointer entry point (frames in, flow out)
    void* hstage = nullptr;
    size_t hstage_bytes = 0;
    // row-filtered intermediate of the two-pass pyramid kernels
    void* tmp = nullptr;
    size_t tmp_bytes = 0;
    // work-list path: per-level item tables (pinned host copy + device copy, two slots used alternately) and the
    // events that mark the end of each slot's last upload (the pinned copy is rewritten two calls later)
    void* het_h = nullptr;
    void* het_d = nullptr;
    size_t het_bytes = 0;
    hipEvent_t het_ev[2] = {nullptr, nullptr};
    int het_flip = 0;
    // pipelined host entry (nsof_farneback_u8_batch): copy streams, per-slot staging and events
    struct nsof_pipe* pipe = nullptr;
    // level overlap of the uniform batch driver: a side stream for the LDS-free stages (pyramid level of the next
    // level, flow resample) that share the CUs with the LDS-bound iteration / expansion kernels, and its events
    hipStream_t side = nullptr;
    std::vector<hipEvent_t> ov_events;
    // exact-order fused iteration (farneback_iterate_x.hip): strip-to-strip carries (tagged granules, zeroed when
    // allocated, never again: a launch's tag is its epoch), the per-XCD ticket counters + timeout word (x_sync:
    // tickets at word 0, timeout word at word 256), the launch epoch, and whether a launch's timeout word needs a look
    // private flow buffers of ROI crops that overlap an earlier crop of the same frame pair (nsof_farneback_u8_roi_sequence_dev)
    void* roi_tmp = nullptr;
    size_t roi_tmp_bytes = 0;
    // ... and the table of their ordered pastes (pinned host copy, device copy, the event after its last upload)
    void* paste_h = nullptr;
    void* paste_d = nullptr;
    size_t paste_bytes = 0;
    hipEvent_t paste_ev = nullptr;
    unsigned long long* x_carry = nullptr;
    size_t x_carry_bytes = 0;
    unsigned* x_sync = nullptr;
    unsigned x_epoch = 0;
    bool x_dirty = false;
};

// Tuning / A-B switches.  Only builds made by scripts/build_variant.sh (-DNSOF_AB) read them; in the product library each
// of them is the constant "unset" and the larger kernels they select are not compiled (#ifdef NSOF_AB in the sources).
// The environment variables the PRODUCT reads are the context defaults documented in include/nsof.h (NSOF_POLYEXP_F32,
// NSOF_EXACT_ROWSUMS, NSOF_PYR_FMA, NSOF_LAT_JOBS, NSOF_ROW_BANDS), the pipelined entry's NSOF_PIPE_CHUNK_MB /
// NSOF_PIPE_FAIL_AFTER_CHUNK / NSOF_PIPE_TRACE and the chunking cap NSOF_MAX_PAIRS (test hooks).
#ifdef NSOF_AB
#define NSOF_AB_GETENV(name) getenv(name)
#else
#define NSOF_AB_GETENV(name) (static_cast<const char*>(nullptr))
#endif

int nsof_set_error(nsof_ctx* ctx, int code, const char* fmt, ...);
int nsof_ws_reserve(nsof_ctx* ctx, void** buf, size_t* cur, size_t need);
// Page-locked host memory on the GPU's NUMA node (best effort); NUMA node of a device from sysfs, -1 if unknown.
void* nsof_pinned_alloc(int device, size_t bytes);
int nsof_gpu_numa_node(int device);
// Grow ctx->hstage (pinned host staging) to at least `need` bytes.
int nsof_hstage_reserve(nsof_ctx* ctx, size_t need);

#define NSOF_HIP(ctx, call)                                                                      \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return nsof_set_error((ctx), NSOF_EDEVICE, "%s failed: %s (%s:%d)", #call,           \
                                  hipGetErrorString(e_), __FILE__, __LINE__);                    \
    } while (0)

// Event bracketing of one launch when profiling of kernel `id` is enabled.
struct nsof_prof_scope {
    nsof_ctx* ctx;
    int id;
    bool on;
    nsof_prof_scope(nsof_ctx* c, int k);
    ~nsof_prof_scope();
};

// ---- Farneback driver pieces shared between nsof_api.hip and farneback_batch.hip -------------------------------
int nsof_check_farneback_params(nsof_ctx* ctx, int width, int height, double pyr_scale, int levels, int winsize,
                                int iterations, int poly_n, int flags);
// Uniform-shape device batch (sequence == true: n_pairs + 1 consecutive frames in d_prev).
int nsof_farneback_core(nsof_ctx* ctx, bool sequence, int n_pairs, const uint8_t* d_prev, const uint8_t* d_next,
                        ptrdiff_t row_stride, ptrdiff_t pair_stride, int width, int height, float* d_flow,
                        double pyr_scale, int levels, int winsize, int iterations, int poly_n, double poly_sigma,
                        int flags);
void nsof_pipe_destroy(nsof_ctx* ctx);

// ---- Farneback launchers (farneback_kernels.hip) ------------------------------------------
struct nsof_blur_taps {
    int ksize;
    float k[NSOF_MAX_BLUR_TAPS];
};
struct nsof_poly_taps {
    int n;
    float g[NSOF_MAX_POLY_N + 1], xg[NSOF_MAX_POLY_N + 1], xxg[NSOF_MAX_POLY_N + 1];
    double dg[NSOF_MAX_POLY_N + 1], dxxg[NSOF_MAX_POLY_N + 1];
    double ig11, ig03, ig33, ig55;
};

int nsof_host_blur_taps(int ksize, double sigma, nsof_blur_taps* out);
int nsof_host_poly_taps(int n, double sigma, nsof_poly_taps* out);

// ---- shape-heterogeneous work lists (nsof_farneback_u8_batch*) ------------------------------------------------
// One work item (a frame pair of its own shape) at ONE pyramid level.  The host builds one table per level (items
// that have no such level are left out) and every stage is launched once per level over the whole table:
// gridDim.z indexes the table (x2 for the per-image stages), gridDim.x/y are sized for the largest item and the
// workgroups outside an item's extent leave at once.  Offsets are element offsets into the level's workspace
// buffers: I (level images, f32; prev at offI, next at offI + wk*hk), R (expansions, f32; R0 at offR, R1 at
// offR + 5*wk*hk), flow (float2; this level at offF, the coarser level's field at offFc).
struct nsof_het_item {
    const uint8_t* src[2];       // full-resolution u8 frames (prev, next), device memory
    long long src_stride[2];     // their row strides in bytes
    float* out;                  // the caller's flow field of this item (written by the last iteration of level 0)
    long long out_pitch;         // its row pitch in float2 units
    unsigned long long offI, offR, offF, offFc;
    int W, H;                    // full resolution
    int wk, hk;                  // this level
    int pw, ph;                  // coarser level (0: the item starts here, its incoming flow is zero)
    int flags;                   // NSOF_HET_VEC0: both frames 4-byte aligned with W % 4 == 0 (vector level-0 kernel)
    int pad_;
};
enum { NSOF_HET_VEC0 = 1 };

// Small-batch exact-order iteration (farneback_iterate_lat.hip): matrices, column sums and row scan as three wide kernels.
// M: 5 floats, V: 5 doubles per pixel of the level (work list: at offR / 2 of each item).  winsize 2..15.
int nsof_launch_iterate_lat(nsof_ctx* ctx, int n_pairs, const float* R0, const float* R1, size_t pair_stride,
                            const float* flow_in, float* flow_out, int W, int H, int winsize, float* M, double* V);
int nsof_launch_iterate_lat_het(nsof_ctx* ctx, int n_items, const nsof_het_item* d_items, int max_w, int max_h, const float* R,
                                const float* flow_in, float* flow_out, bool final, int winsize, float* M, double* V);
int nsof_launch_rowscan_solve(nsof_ctx* ctx, int n_pairs, const double* V, int W, int H, int winsize, float* flow_out);
int nsof_launch_rowscan_solve_het(nsof_ctx* ctx, int n_items, const nsof_het_item* items, int max_h, const double* V,
                                  float* flow_out, bool final, int winsize);
int nsof_launch_iterate_het_exact(nsof_ctx* ctx, int n_items, const nsof_het_item* d_items, int max_w, int max_h,
                                  const float* R, const float* flow_in, float* flow_out, bool final, int winsize,
                                  double* vsum);
// All launchers are asynchronous on ctx->stream and return an nsof_status.
// The *_het twins take a device table of n_items entries; max_* are the largest extents over the table.
int nsof_launch_prep_het(nsof_ctx* ctx, int n_items, const nsof_het_item* d_items, const nsof_het_item* h_items,
                         bool level0, const nsof_blur_taps& taps, float* I);
int nsof_launch_polyexp_het(nsof_ctx* ctx, int n_items, const nsof_het_item* d_items, int max_w, int max_h,
                            const nsof_poly_taps& taps, const float* I, float* R, const float* blur3 = nullptr);
int nsof_launch_flow_upsample_het(nsof_ctx* ctx, int n_items, const nsof_het_item* d_items, int max_w, int max_h,
                                  const float* src, float* dst, float mul);
// final: the flow goes to the items' own output fields (out / out_pitch) instead of flow_out.
int nsof_launch_iterate_het(nsof_ctx* ctx, int n_items, const nsof_het_item* d_items, int max_w, const float* R,
                            const float* flow_in, float* flow_out, bool final, int winsize);
int nsof_launch_prep(nsof_ctx* ctx, int n_img, const uint8_t* src, ptrdiff_t row_stride, ptrdiff_t img_stride, int W,
                     int H, int wk, int hk, const nsof_blur_taps& taps, float* out);
// The *_fma twins (farneback_kernels.hip compiled with -DNSOF_PYR_FMA): same taps and order, every tap / blend one fused
// multiply-add -- selected by ctx->opt_pyr_fma through the *_sel wrappers below.
int nsof_launch_prep_fma(nsof_ctx* ctx, int n_img, const uint8_t* src, ptrdiff_t row_stride, ptrdiff_t img_stride, int W,
                         int H, int wk, int hk, const nsof_blur_taps& taps, float* out);
// Levels 1..3 of a pyr_scale 0.5 pyramid in one launch; NSOF_EUNSUPPORTED (nothing launched) when the frames do not qualify.
int nsof_launch_prep_decim3(nsof_ctx* ctx, int n_img, const uint8_t* src, ptrdiff_t row_stride, ptrdiff_t img_stride, int W,
                            int H, const nsof_blur_taps* taps, float* const* out);
int nsof_launch_prep_decim3_fma(nsof_ctx* ctx, int n_img, const uint8_t* src, ptrdiff_t row_stride, ptrdiff_t img_stride,
                                int W, int H, const nsof_blur_taps* taps, float* const* out);
int nsof_launch_prep_het_fma(nsof_ctx* ctx, int n_items, const nsof_het_item* d_items, const nsof_het_item* h_items,
                             bool level0, const nsof_blur_taps& taps, float* I);
int nsof_launch_flow_upsample_fma(nsof_ctx* ctx, int n_pairs, const float* src, int sw, int sh, float* dst, int dw,
                                  int dh, float mul);
int nsof_launch_flow_upsample_het_fma(nsof_ctx* ctx, int n_items, const nsof_het_item* d_items, int max_w, int max_h,
                                      const float* src, float* dst, float mul);
int nsof_launch_polyexp(nsof_ctx* ctx, int n_img, const float* img, int W, int H, const nsof_poly_taps& taps,
                        float* R);
// Full-resolution level: pyramid level (3-tap smoothing, centre k0 / side k1) + expansion in one kernel, from the frames.
int nsof_launch_polyexp_u8(nsof_ctx* ctx, int n_img, const uint8_t* src0, const uint8_t* src1, int nsplit, ptrdiff_t row_stride,
                           ptrdiff_t img_stride, int W, int H, const nsof_poly_taps& taps, float k0, float k1, float* R);
// R0/R1: planar [5][h][w] expansion of prev/next of pair 0; pair z is at +z*pair_stride floats.
int nsof_launch_update_matrices(nsof_ctx* ctx, int n_pairs, const float* R0, const float* R1, size_t pair_stride,
                                const float* flow, int W, int H, float* M);
int nsof_launch_blur_solve(nsof_ctx* ctx, int n_pairs, const float* M, int W, int H, int winsize, float* flow);
// The same in the reference library's exact summation order; VT: n_pairs * 5 * W * H doubles of scratch.
int nsof_launch_blur_solve_exact(nsof_ctx* ctx, int n_pairs, const float* M, int W, int H, int winsize, double* VT,
                                 float* flow);
int nsof_launch_flow_upsample(nsof_ctx* ctx, int n_pairs, const float* src, int sw, int sh, float* dst, int dw,
                              int dh, float mul);
#define NSOF_PYR_SEL(ctx, fn, ...) ((ctx)->opt_pyr_fma ? fn##_fma(ctx, __VA_ARGS__) : fn(ctx, __VA_ARGS__))
bool nsof_iterate_supported(int winsize, int W, int H);
// Fused iteration; flow_in != flow_out.
int nsof_launch_iterate(nsof_ctx* ctx, int n_pairs, const float* R0, const float* R1, size_t pair_stride,
                        const float* flow_in, float* flow_out, int W, int H, int winsize);
// Exact-order twin (row sums as one running sum per image row, the reference library's order): vsum = n_pairs * 5 * W * H
// doubles of scratch (the column sums pass through HBM between its two kernels).
bool nsof_iterate_exact_supported(int winsize, int W, int H);
int nsof_launch_iterate_exact(nsof_ctx* ctx, int n_pairs, const float* R0, const float* R1, size_t pair_stride,
                              const float* flow_in, float* flow_out, int W, int H, int winsize, double* vsum);
// Exact-order fused iteration in ONE kernel (running row sums inside the strip walker, strips chained by carries).
bool nsof_iterate_x_supported(int winsize, int W, int H);
int nsof_launch_iterate_x(nsof_ctx* ctx, int n_pairs, const float* R0, const float* R1, size_t pair_stride,
                          const float* flow_in, float* flow_out, int W, int H, int winsize);
// d_xjobs: the level's job table -- 8 counts, then 8 lists (one per XCD, `stride` entries apart) of item << 8 | strip;
// njobs = the sum of the counts = the grid; strips are NSOF_X_STRIP output columns wide.
#define NSOF_X_STRIP 192
int nsof_launch_iterate_x_het(nsof_ctx* ctx, int n_items, const nsof_het_item* d_items, int max_w, int max_h, const float* R,
                              size_t R_floats, const float* flow_in, float* flow_out, bool final, int winsize,
                              const unsigned* d_xjobs, int stride, int njobs);
// Carry buffer of at least `carry_bytes` (0: whatever exists) + ticket / timeout words of that kernel.
int nsof_xsync_reserve(nsof_ctx* ctx, size_t carry_bytes, unsigned long long** carry, unsigned** tickets, unsigned** err);
// Reads the timeout word of the exact-order kernel after the stream has drained; NSOF_EDEVICE if a carry never arrived.
int nsof_xsync_check(nsof_ctx* ctx);
// hipStreamSynchronize(ctx->stream) + nsof_xsync_check: the tail of every entry point that hands results to the host
int nsof_stream_sync_checked(nsof_ctx* ctx);
bool nsof_iterate_upsample_supported(int winsize, int W, int H);
// First iteration of a level: flow_in = resample(coarse_flow [sh][sw][2]) * mul, computed on the fly.
int nsof_launch_iterate_upsample(nsof_ctx* ctx, int n_pairs, const float* R0, const float* R1, size_t pair_stride,
                                 const float* coarse_flow, int sw, int sh, float mul, float* flow_out, int W, int H,
                                 int winsize);

// ---- streamed stores ----------------------------------------------------------------------------------------
// Outputs that are not re-read before the caches have turned over (pyramid images, R, flow fields of a batch) are
// written with non-temporal stores: measured on MI355X, the flow resample kernel goes from 3.5 to 6.2 TB/s and the
// level-0 pyramid kernel from 3.7 to 4.4 TB/s (plain stores write-allocate in L2 and evict what the gathers reuse).
// NSOF_PLAIN_STORES builds the plain-store variant for A/B runs.
#ifdef __HIPCC__
typedef float nsof_f4v __attribute__((ext_vector_type(4)));
typedef float nsof_f2v __attribute__((ext_vector_type(2)));
// 1/x, correctly rounded, for a NORMAL x whose reciprocal is normal too (the determinant + 1e-3 of the 2x2 systems: between
// ~1e-3 and ~1e20 for 8-bit frames).  This is the division sequence the compiler emits for 1./x -- v_rcp_f64, two Newton
// steps, the residual correction of the quotient -- without the operand scaling and special-case fix-ups
// (v_div_scale / v_div_fmas / v_div_fixup: 5 of the 12 instructions) that only matter for subnormal, huge or non-finite
// operands.  Bit-identical to IEEE division on that range (tests/test_farneback_gpu.py::test_recip_matches_ieee_division).
__device__ __forceinline__ double nsof_recip_normal(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-x, r, 1.0);   // residual of the quotient q0 = 1 * r
    return __builtin_fma(e, r, r);
}

__device__ __forceinline__ void nsof_store_stream4(float* p, float a, float b, float c, float d)
{
#ifndef NSOF_PLAIN_STORES
    __builtin_nontemporal_store((nsof_f4v){a, b, c, d}, reinterpret_cast<nsof_f4v*>(p));
#else
    *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
#endif
}
// (Non-temporal LOADS of the once-read inputs of the iteration kernel were measured too: 3 % slower.)
__device__ __forceinline__ void nsof_store_stream2(float* p, float a, float b)
{
#ifndef NSOF_PLAIN_STORES
    __builtin_nontemporal_store((nsof_f2v){a, b}, reinterpret_cast<nsof_f2v*>(p));
#else
    *reinterpret_cast<float2*>(p) = make_float2(a, b);
#endif
}
#endif
