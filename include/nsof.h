/*
 * nsof.h -- C ABI of libnsof.so, the MI355X (gfx950) neuromorphic optical-flow core.
 *
 * This is the drop-in boundary for the two data-parallel stages of
 * RTCartist/Neuromorphic-Spatiotemporal-Optical-Flow (paths below are relative to the
 * reference checkout):
 *
 *   stage 2  dense Farneback flow -- replaces `cv2.calcOpticalFlowFarneback(prev, next,
 *            None, pyr_scale, levels, winsize, iterations, poly_n, poly_sigma, flags)`
 *            called at optical_flow_seg.py:158,203,494, optical_flow_ob.py:225,270,616,
 *            optical_flow_prediction.py:161,206,570, optical_flow_yolo.py:199,244,898
 *            (parameter dict: optical_flow_seg.py:73-81).
 *   stage 1  synaptic accumulator -- replaces update_state / resistance_exp / simulate of
 *            eventsim/event_mem_sim.py:40-57, :60-63, :164-286.
 *
 * Plain C types only; no torch / numpy types cross this boundary.  All entry points
 * return NSOF_OK (0) or a negative nsof_status; nsof_last_error() gives the text.
 * A context owns one device, one HIP stream and a reusable workspace; it is not
 * thread-safe, several contexts may coexist.  There is no CPU fallback: without a
 * usable gfx950 device nsof_create() fails with NSOF_EDEVICE.
 */
#ifndef NSOF_H
#define NSOF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NSOF_ABI_VERSION 3   /* 2: NSOF_OPT_EXACT_ROWSUMS defaults to 1; option ranges validated.  3: gate_frame argument of
                              * nsof_farneback_u8_roi_sequence_dev, nsof_accum_set_slice_times, NSOF_OPT_DEBUG_FAULT; a lost
                              * hand-over fails the synchronising call */

typedef enum nsof_status {
    NSOF_OK = 0,
    NSOF_EINVAL = -1,       /* bad argument (cv2 would raise cv2.error on its CV_Assert) */
    NSOF_ESHAPE = -2,       /* prev/next shapes differ or are empty */
    NSOF_EDEVICE = -3,      /* HIP runtime / device failure, or no gfx950 device */
    NSOF_ENOMEM = -4,       /* host or device allocation failed */
    NSOF_EUNSUPPORTED = -5  /* flags the reference never uses (USE_INITIAL_FLOW=4, FARNEBACK_GAUSSIAN=256) */
} nsof_status;

typedef struct nsof_ctx nsof_ctx;

/* ---- context ------------------------------------------------------------------------- */
/* device: HIP device ordinal.  *out receives the context. */
int nsof_create(int device, nsof_ctx** out);
void nsof_destroy(nsof_ctx* ctx);
/* Text of the last error on this context (or of the last failed nsof_create if ctx==NULL). */
const char* nsof_last_error(const nsof_ctx* ctx);
int nsof_abi_version(void);
/* Launch on a caller-owned hipStream_t (e.g. torch's current stream); NULL -> own stream. */
int nsof_set_stream(nsof_ctx* ctx, void* hip_stream);
int nsof_synchronize(nsof_ctx* ctx);
/* Options.  NSOF_OPT_POLYEXP_F32 (default 0): 1 = the polynomial-expansion kernel accumulates its horizontal
 * moments in float instead of double -- faster, but NOT the reference library's arithmetic: the flow then differs
 * from the default (exact) path by the end-point error DESIGN.md reports.  Applies to the uniform-shape entry points;
 * the work-list entries always run the exact kernel.  New contexts take the default from the environment
 * variable NSOF_POLYEXP_F32. */
/* NSOF_OPT_EXACT_ROWSUMS (default 1): the box-filter row sums are formed as the reference library forms them, ONE
 * running double-precision sum along each image row (g += vsum[x+m] - vsum[x-m-1]), inside the fused iteration kernel:
 * every stage then keeps the library's operation order and the flow equals a CPU restatement of the library bit for bit
 * on any input (windows up to 15: one kernel per iteration; larger windows: the unfused kernels, same results).
 * 0 = the "fast row sums" mode: each pixel's window is summed directly -- the same numbers to ~1e-16, a few per cent
 * faster; where the 2x2 system is rank deficient (straight edges and flat areas of real footage, small windows) the
 * rounding history decides the flow's 4th decimal and this mode leaves the library by up to ~8e-4.
 * Environment default: NSOF_EXACT_ROWSUMS. */
/* NSOF_OPT_ROW_BANDS (default 0; applies to the fast row-sum mode only, NSOF_OPT_EXACT_ROWSUMS = 0): for SMALL batches
 * (one call per camera frame, the reference's own call pattern).  The
 * fused iteration kernel walks an image strip top to bottom in one workgroup, because the library's column sums are one
 * running sum from row 0; a lone 1080p pair then occupies 8 of 256 compute units.  1 = split every strip into row bands
 * (automatic height; applied from winsize 9 up), >= 4 = bands of that many rows at any window: each band starts its
 * column sums with a direct sum of its first window, which lacks the rounding history of the running sum -- the same
 * class of deviation as the row-sum order above but more frequent: ~1e-5 on textured frames with wide windows, 4th
 * decimal at many pixels with 3x3 / 4x4 windows (docs/HISTORY_r1_r3.md section 5.1 has the soak counts).  Off by default so that a
 * pair's flow does not depend on the batch it was part of.  Values 2 and 3 are rejected.  Environment default:
 * NSOF_ROW_BANDS. */
/* NSOF_OPT_PYR_FMA (default 0): the arithmetic-variant twin of the pyramid stages.  0 = the float Gaussian blur and the
 * bilinear resamples (pyramid levels, flow resize between levels) round every product and every sum, as the library's
 * generic C++ code does; 1 = the same taps in the same order with one fused multiply-add per tap / blend, as an AVX2+FMA3
 * build of the library's vector loops contracts them.  Which of the two a given cv2 wheel executes cannot be pinned in
 * this image (DESIGN.md section 2 states how far the flow moves between them); both equal their CPU restatement
 * (oracle build of the same switch) bit for bit.  Environment default: NSOF_PYR_FMA. */
/* NSOF_OPT_SMALL_BATCH_JOBS (default 64; exact row-sum order only): the fused iteration kernel gives one workgroup a
 * (192-column strip, image) job and walks the image height in it; a call with at most this many jobs -- a lone frame
 * pair (10 jobs at 1920 wide), a few ROI crops: the reference's own call pattern -- leaves most of the 256 compute
 * units idle, so such calls run the SAME arithmetic in the SAME order as three wide kernels per iteration (matrices;
 * column sums, one thread per column and plane; row sums + solve, one thread per row and plane) with the intermediates
 * in HBM: a lone 1920x1080 call 3.9 -> 1.2 ms host to host.  Results are bit-identical either way; the value only moves
 * the switch-over (measured cross-over: 70-80 jobs, scripts/small_batch_crossover.py; 0 = always the fused kernel).
 * Environment default: NSOF_LAT_JOBS. */
/* NSOF_OPT_DEBUG_FAULT (default 0; TEST HOOK, not a mode): the exact-order iteration kernels hand running sums from one
 * workgroup / wave to the next (strip-to-strip carries in k_iterate_x, wave turns in k_lat_colsum); every wait for such a
 * hand-over is bounded, and a wait that runs out makes the call that synchronises next return NSOF_EDEVICE (cv2 raises
 * cv2.error where it fails, optical_flow_seg.py:203 -- a flow field is never handed back from a failed launch).  Bit 0 = strip
 * 0 of item 0 of every k_iterate_x launch withholds its carries, bit 1 = wave 3 of workgroup (0, 0, 0) of every
 * k_lat_colsum launch never passes its turn on: the failure the bounded waits exist for, on demand
 * (tests/test_farneback_gpu.py::test_lost_handover_is_reported). */
enum { NSOF_OPT_POLYEXP_F32 = 1, NSOF_OPT_EXACT_ROWSUMS = 2, NSOF_OPT_ROW_BANDS = 3, NSOF_OPT_PYR_FMA = 4,
       NSOF_OPT_SMALL_BATCH_JOBS = 5, NSOF_OPT_DEBUG_FAULT = 100 };
int nsof_set_option(nsof_ctx* ctx, int option, int value);
int nsof_get_option(const nsof_ctx* ctx, int option, int* value);

/* ---- stage 2: Farneback --------------------------------------------------------------- */
/* Same argument meaning and order as cv2.calcOpticalFlowFarneback (see call sites above).
 * prev/next: HOST uint8 single-channel images, row strides in bytes (strided ROI views are
 * fine, optical_flow_seg.py:186-187); flow: HOST float32, interleaved (u,v), row stride in
 * bytes.  Pointers are not retained.  Blocks until the flow is in host memory. */
int nsof_farneback_u8(nsof_ctx* ctx,
                      const uint8_t* prev, ptrdiff_t prev_stride,
                      const uint8_t* next, ptrdiff_t next_stride,
                      int width, int height,
                      float* flow, ptrdiff_t flow_stride,
                      double pyr_scale, int levels, int winsize, int iterations,
                      int poly_n, double poly_sigma, int flags);

/* Batched, device-resident twin: n_pairs frame pairs of one shape already in HBM.
 * d_prev/d_next: DEVICE uint8 [n_pairs][height][row_stride]; pair_stride in bytes.
 * d_flow: DEVICE float32 [n_pairs][height][width][2] (dense).  Asynchronous on the
 * context's stream. */
int nsof_farneback_u8_batch_dev(nsof_ctx* ctx, int n_pairs,
                                const uint8_t* d_prev, const uint8_t* d_next,
                                ptrdiff_t row_stride, ptrdiff_t pair_stride,
                                int width, int height, float* d_flow,
                                double pyr_scale, int levels, int winsize, int iterations,
                                int poly_n, double poly_sigma, int flags);

/* Sequence twin: n_frames consecutive frames already in HBM, d_frames uint8 [n_frames][height][row_stride];
 * d_flow float32 [n_frames-1][height][width][2], flow i = (frame i -> frame i+1), exactly what n_frames-1 calls
 * of nsof_farneback_u8 on consecutive frames give (the reference's seg/ob/prediction loops walk a sequence this
 * way, optical_flow_seg.py:413-496).  Each frame's pyramid levels and polynomial expansion are computed once and
 * shared by the two pairs the frame belongs to.  Asynchronous on the context's stream. */
int nsof_farneback_u8_sequence_dev(nsof_ctx* ctx, int n_frames, const uint8_t* d_frames,
                                   ptrdiff_t row_stride, ptrdiff_t frame_stride,
                                   int width, int height, float* d_flow,
                                   double pyr_scale, int levels, int winsize, int iterations,
                                   int poly_n, double poly_sigma, int flags);

/* One frame pair of a shape-heterogeneous batch: what ONE call of cv2.calcOpticalFlowFarneback(prev_region,
 * next_region, None, **farneback_params) receives and returns in the gated path -- optical_flow_seg.py:129-164
 * (one crop per connected component), :186-203 (the union box), :492-496 (the full frame) -- as plain pointers.
 * flow_stride in bytes (a multiple of 8): a view flow_canvas[y0:y1, x0:x1] of a frame-sized float32 (H,W,2)
 * canvas is written in place, which is the paste of :162 / :204. */
typedef struct nsof_pair_desc {
    const uint8_t* prev;
    ptrdiff_t prev_stride;
    const uint8_t* next;
    ptrdiff_t next_stride;
    int width, height;
    float* flow;
    ptrdiff_t flow_stride;
} nsof_pair_desc;

/* n_pairs pairs of ANY shapes, one parameter set, HOST memory (pointers are not retained; blocks until every flow
 * field is in host memory).  All pairs share every kernel launch (a work list per pyramid level), so many small ROI
 * calls cost about as much as one; the list is processed in chunks whose upload, compute and download overlap on
 * three streams.  Buffers from nsof_host_alloc() (page-locked) are copied to/from directly, other memory goes
 * through an internal pinned staging buffer.  Result per pair == nsof_farneback_u8 of that pair, bit for bit, in the default
 * mode; with the opt-in modes that depend on the batch (NSOF_OPT_ROW_BANDS in automatic mode picks its band height from the
 * batch size; a uniform list takes the uniform driver, where NSOF_OPT_POLYEXP_F32 applies) only to their tolerance. */
int nsof_farneback_u8_batch(nsof_ctx* ctx, int n_pairs, const nsof_pair_desc* pairs,
                            double pyr_scale, int levels, int winsize, int iterations,
                            int poly_n, double poly_sigma, int flags);
/* Device-resident twin: `pairs` is a HOST array whose prev/next/flow are DEVICE addresses (e.g. crops of frames
 * and of flow canvases already in HBM).  Asynchronous on the context's stream. */
int nsof_farneback_u8_batch_desc_dev(nsof_ctx* ctx, int n_pairs, const nsof_pair_desc* pairs,
                                     double pyr_scale, int levels, int winsize, int iterations,
                                     int poly_n, double poly_sigma, int flags);
/* The gated path of a whole frame sequence on the device (opticalFlow3D's crop -> flow -> paste loop,
 * /root/reference/optical_flow_seg.py:129-164, 186-204): d_frames = n_frames 8-bit frames in HBM, d_counts / d_rects = the
 * ROI table nsof_roi_from_surface_dev wrote (rects [n_frames][max_rects][4] = x0, y0, x1, y1), d_flows =
 * [n_frames - 1][height][width][2] float canvases, zero-filled here.  Pair k = (frame k, frame k + 1) is gated by the
 * rectangles of frame k + gate_frame: 1 = the map of the pair's second frame (what opticalFlow3D is written to use,
 * memimg2), 0 = the map of its first frame (what the shipped scripts pass: memimg2 := memimg1, optical_flow_seg.py:435 --
 * the bug-compatible default of nsof.gating); every crop of every pair is one item of ONE work list, written into the canvas in place;
 * a crop that overlaps an earlier crop of its pair is pasted after it, in label order, as the reference's loop
 * overwrites.  Each crop's flow equals nsof_farneback_u8 of that crop bit for bit.  Only the rectangle table crosses
 * PCIe (the work list's shapes are needed on the host; the call synchronises the stream once for it).  n_calls /
 * n_pixels (optional) receive the number of crops and their total area.  Asynchronous otherwise. */
int nsof_farneback_u8_roi_sequence_dev(nsof_ctx* ctx, int n_frames, const uint8_t* d_frames, ptrdiff_t row_stride,
                                       ptrdiff_t frame_stride, int width, int height, const int32_t* d_counts,
                                       const int32_t* d_rects, int max_rects, float* d_flows, double pyr_scale, int levels,
                                       int winsize, int iterations, int poly_n, double poly_sigma, int flags,
                                       int gate_frame, long long* n_calls, long long* n_pixels);
/* Page-locked host memory for frames / flow fields handed to nsof_farneback_u8_batch (NULL on failure). */
void* nsof_host_alloc(size_t bytes);
void nsof_host_free(void* p);

/* Geometry helpers (pure host arithmetic, usable without a device: ctx may be NULL). */
int nsof_farneback_effective_levels(int width, int height, double pyr_scale, int levels);
int nsof_farneback_level_size(int width, int height, double pyr_scale, int level,
                              int* level_width, int* level_height, int* blur_ksize, double* blur_sigma);

/* Individual pipeline stages on DEVICE buffers, exposed for stage-level parity tests and
 * for the roofline benchmark.  Layouts: images [n_img][h][w] float32; flow [n_img][h][w][2]
 * float32; M planar [n_img][5][h][w] float32; the polynomial expansion R of ONE image is
 * 5*h*w float32 = [h][w][4] (channels 0..3 interleaved per pixel) followed by [h][w]
 * (channel 4), images back to back. */
int nsof_stage_pyr_level(nsof_ctx* ctx, int n_img, const uint8_t* d_src, ptrdiff_t row_stride,
                         ptrdiff_t img_stride, int width, int height, double pyr_scale, int level, float* d_out);
int nsof_stage_polyexp(nsof_ctx* ctx, int n_img, const float* d_img, int width, int height,
                       int poly_n, double poly_sigma, float* d_R);
/* Diagnostic: d_out[i] = the reciprocal the 2x2 solves use (rcp + Newton steps + residual correction, without the
 * scaling / fix-up instructions of a general division) and d_ieee[i] = 1.0 / d_x[i] as the compiler divides; the test
 * asserts they are the same bits over the determinants' range. */
int nsof_stage_recip(nsof_ctx* ctx, long long n, const double* d_x, double* d_out, double* d_ieee);
/* n_pairs pairs: R holds [n_pairs][2][5][h][w] (image 0 = prev, 1 = next). */
int nsof_stage_update_matrices(nsof_ctx* ctx, int n_pairs, const float* d_R, const float* d_flow,
                               int width, int height, float* d_M);
int nsof_stage_blur_solve(nsof_ctx* ctx, int n_pairs, const float* d_M, int width, int height,
                          int winsize, float* d_flow);
/* One fused Farneback iteration (matrix update + blur + solve): flow_out = step(R, flow_in).  d_flow_in and
 * d_flow_out must not alias.  winsize 2..17; larger windows take the unfused pair above. */
int nsof_stage_iterate(nsof_ctx* ctx, int n_pairs, const float* d_R, const float* d_flow_in, int width, int height,
                       int winsize, float* d_flow_out);
/* The same with flow_in = resample(d_coarse_flow [src_h][src_w][2]) * (1/pyr_scale) formed on the fly (first
 * iteration of a pyramid level): equals nsof_stage_flow_upsample followed by nsof_stage_iterate, bit for bit. */
int nsof_stage_iterate_upsample(nsof_ctx* ctx, int n_pairs, const float* d_R, const float* d_coarse_flow,
                                int src_w, int src_h, int width, int height, int winsize, double pyr_scale,
                                float* d_flow_out);
int nsof_stage_flow_upsample(nsof_ctx* ctx, int n_pairs, const float* d_src, int src_w, int src_h,
                             float* d_dst, int dst_w, int dst_h, double pyr_scale);

/* ---- profiling (HIP events on the context's stream, around every launch of one kernel) -- */
typedef enum nsof_kernel_id {
    NSOF_K_PREP = 0,        /* u8 -> f32, Gaussian blur, bilinear resample (per level) */
    NSOF_K_POLYEXP = 1,     /* polynomial expansion */
    NSOF_K_UPSAMPLE = 2,    /* coarse-to-fine flow resample */
    NSOF_K_UPDMAT = 3,      /* matrix update with warped R1 */
    NSOF_K_BLUR = 4,        /* box blur + 2x2 solve */
    NSOF_K_ACCUM = 5,       /* accumulator state update */
    NSOF_K_ITERATE = 6,     /* fused matrix update + box blur + solve (one Farneback iteration) */
    NSOF_K_SEGMENT = 7,     /* |flow| > th (or u8 != 0) -> bit-packed mask */
    NSOF_K_MORPH = 8,       /* fused dilate/erode chain on the bit-packed mask */
    NSOF_K_REMAP = 9,       /* 8-bit bilinear remap / fused prediction warp */
    NSOF_K_SSIM = 10,       /* SSIM window statistics + per-block partial sums */
    NSOF_K_COUNT = 11
} nsof_kernel_id;
/* mask: bit (1<<id) enables event bracketing for that kernel; 0 disables. */
int nsof_prof_enable(nsof_ctx* ctx, unsigned mask);
/* Synchronises, then returns summed device time (ms) and launch count since the last call. */
int nsof_prof_collect(nsof_ctx* ctx, int kernel_id, double* total_ms, long long* launches);
const char* nsof_kernel_name(int kernel_id);

/* ---- stage 1: synaptic accumulator ---------------------------------------------------- */
typedef struct nsof_accum nsof_accum;

/* Device constants follow event_mem_sim.py:20-34 (PARAMS, DT=5e-4, THETA_EVENTS=1,
 * REFRACTORY_US=800).  scheme: 1 = boxcar window, 2 = DC bias + event overlay (:208-286).
 * polarity_split: scheme 2 only -- 1 = 'split' (ON p==1 -> array A, OFF p==0 -> array B),
 * 0 = 'magnitude' (one array). */
int nsof_accum_create(nsof_ctx* ctx, int height, int width, int scheme, int polarity_split,
                      float active_v, float silent_v, nsof_accum** out);
void nsof_accum_destroy(nsof_accum* acc);
/* With silent_v in the dead zone an idle pixel is a bit-exact no-op, so two updates give the same state: the event-pixel
 * update (only touched pixels, groups of 32 slices) and the every-pixel pass (scheme 1: groups of 64 slices, no lists).
 * force_dense 0 (default) = automatic: scheme 1 takes the every-pixel pass up to ~12 M pixels (it needs half the launches:
 * 1280x720 0.39 vs 0.75-0.89 ms per 30 surface frames, 3840x2160 0.97 vs 1.21 ms), the event-pixel update beyond and in
 * scheme 2; > 0 = always the every-pixel pass (the roofline run); < 0 = the event-pixel update wherever it is exact. */
int nsof_accum_set_dense(nsof_accum* acc, int force_dense);
/* Reset w to wini (0.5) and the refractory maps to 0. */
int nsof_accum_reset(nsof_accum* acc);
/* Advance by n_slices time slices.  Events are HOST arrays (x,y int16; p int8; t int64 us,
 * sorted) as in the /CD/events group (event_mem_sim.py:69-75); slice_bounds has
 * n_slices+1 entries: slice i covers events [slice_bounds[i], slice_bounds[i+1]).  If
 * snap_every > 0, after every slice whose global index (counted since reset) is a multiple
 * of snap_every a resistance snapshot is appended to the snapshot ring. */
int nsof_accum_step_events(nsof_accum* acc, const int16_t* x, const int16_t* y, const int8_t* p,
                           const int64_t* t, const int64_t* slice_bounds, int64_t n_slices,
                           int64_t snap_every);
/* The same in two halves, for streams that are advanced piecewise or re-run: nsof_accum_set_events uploads the
 * events of slices [0, n_slices) ONCE (arrays and bounds as above; nothing is retained but the device copy), and
 * nsof_accum_run advances over any sub-range [first_slice, first_slice + n_slices) of them without touching the
 * host arrays again.  nsof_accum_step_events == set_events followed by run(0, n_slices). */
int nsof_accum_set_events(nsof_accum* acc, const int16_t* x, const int16_t* y, const int8_t* p,
                          const int64_t* t, const int64_t* slice_bounds, int64_t n_slices);
int nsof_accum_run(nsof_accum* acc, int64_t first_slice, int64_t n_slices, int64_t snap_every);
/* Scheme 2 only (a no-op for scheme 1): replace the per-slice timestamps nsof_accum_set_events derived from the staged
 * events -- the first event's time (refractory test) and the last event's (next_ok = t_last + 800 us),
 * event_mem_sim.py:243-267 -- by those of the WHOLE stream's slices.  For an accumulator that holds a row band of the
 * sensor and was staged with the band's events only: its state then equals the band's rows of the unsharded run
 * (SURVEY.md section 8e).  n_slices must equal the staged count; call between set_events and run. */
int nsof_accum_set_slice_times(nsof_accum* acc, const int64_t* t_first, const int64_t* t_last, int64_t n_slices);
/* The current surface as an 8-bit frame on the DEVICE, d_out uint8 [H][row_stride]; asynchronous on the context's
 * stream.  mode NSOF_SURFACE_CURRENT: g = uint8(clip(-3366/log10(I) - 306, 0, 255)), I = 1/R, R = resistance_exp(w) --
 * the reference's map from device current to the gating image (optical_flow_seg.py:426-431) applied per pixel
 * (calibrated for arrays that start at w = 0; the event simulator's initial w = 0.5 already gives 255).
 * mode NSOF_SURFACE_STATE: g = uint8(255 * w), the build-defined frame of the joined events -> flow pipeline. */
enum { NSOF_SURFACE_CURRENT = 0, NSOF_SURFACE_STATE = 1 };
int nsof_accum_surface_u8_dev(nsof_accum* acc, int which, int mode, uint8_t* d_out, ptrdiff_t row_stride);
/* nsof_accum_run + nsof_accum_surface_u8_dev of the state after the run's last slice, as one call: where the dense
 * scheme-1 update runs, its last pass writes the frame itself (no separate pass over the array, one launch less); the
 * frame is byte-identical to the two calls. */
int nsof_accum_run_surface(nsof_accum* a, int64_t first_slice, int64_t n_slices, int which, int mode, uint8_t* d_out,
                           ptrdiff_t row_stride);
/* n_frames consecutive intervals of `every` slices, the surface after each interval into d_frames + k * frame_stride (uint8
 * [H][row_stride] each) -- what n_frames calls of nsof_accum_run_surface give, byte for byte, issued from one call.  Scheme 1
 * with the silent voltage in the dead zone [voff, von] (no forced every-pixel pass, every <= 64): frame k is frame k-1 copied
 * (2 B/px) and patched at the pixels the interval's events touched, the state updated at those pixels only -- or, by
 * default, the whole run as one tile-persistent walk (nsof_accum_set_frames_path).  Asynchronous on the context's stream. */
int nsof_accum_run_frames(nsof_accum* acc, int64_t first_slice, int64_t n_frames, int64_t every, int which, int mode,
                          uint8_t* d_frames, ptrdiff_t row_stride, ptrdiff_t frame_stride);
/* Which of its two event-driven forms nsof_accum_run_frames takes where both apply (same bytes): 0 (default) = the tile
 * walk -- a wave owns 1024 consecutive pixels for the whole run (state, current frame bytes and slice masks in LDS), the
 * run's events are bucketed by (interval, tile) first, frames are write-only: two launches per call; needs sensor width,
 * row and frame strides that are multiples of 16 and a 16-byte aligned frame buffer -- 1 = copy + patch per interval (two
 * launches per interval; what the tile walk falls back to, and its cross-check in the tests). */
int nsof_accum_set_frames_path(nsof_accum* acc, int path);
/* Checkpoint / resume (the reference persists only w_final, event_mem_sim.py:289-303): copy one array's state to /
 * from HOST memory -- w float32 [H][W], the refractory map int64 [H][W] (scheme 2; zeros otherwise) and the global
 * slice counter that times the snapshots.  NULL pointers are skipped. */
int nsof_accum_read_state(nsof_accum* acc, int which, float* w_out, int64_t* next_ok_out, int64_t* slice_counter);
int nsof_accum_write_state(nsof_accum* acc, int which, const float* w_in, const int64_t* next_ok_in,
                           int64_t slice_counter);
/* Dense element-wise update_state on DEVICE arrays (event_mem_sim.py:40-57). */
int nsof_accum_update_state_dev(nsof_ctx* ctx, const float* d_w, const float* d_V, float* d_out, size_t n);
/* bincount_2d(x, y, H, W) of event_mem_sim.py:100-104: events per pixel, int32 [H][W].  HOST arrays in and out;
 * events outside the sensor are an error (np.bincount would raise or grow the array). */
int nsof_accum_bincount_2d(nsof_ctx* ctx, const int16_t* x, const int16_t* y, size_t n, int height, int width,
                           int32_t* counts_out);
/* Dense resistance_exp on DEVICE arrays (event_mem_sim.py:60-63). */
int nsof_accum_resistance_dev(nsof_ctx* ctx, const float* d_w, float* d_out, size_t n);
/* Copy state to HOST: which = 0 (array A) or 1 (array B, split mode). */
int nsof_accum_read_w(nsof_accum* acc, int which, float* w_out);
int nsof_accum_read_resistance(nsof_accum* acc, int which, float* r_out);
/* Snapshots taken so far; copy them ([count][H][W] float32) to HOST and clear the ring. */
int64_t nsof_accum_snapshot_count(const nsof_accum* acc);
int nsof_accum_read_snapshots(nsof_accum* acc, int which, float* out, int64_t max_count);
/* Input of the gating image formed on the device: out[H/memsize][W/memsize] (HOST, float64) = the maximum of the device
 * current v_ds / R over every memsize x memsize block of pixels -- of stored snapshot `snapshot` (0-based, not consumed), or
 * of the current state when snapshot < 0.  Same values as dividing the downloaded float32 resistance map on the host
 * (max of v_ds/R = v_ds / min R); only rows x cols doubles cross PCIe instead of the whole surface. */
int nsof_accum_block_current(nsof_accum* acc, int which, int64_t snapshot, int memsize, double v_ds, double* out);
/* The same map written to DEVICE memory (d_out: rows x cols doubles), asynchronous on the context's stream: the input of
 * nsof_roi_from_surface_dev. */
int nsof_accum_block_current_dev(nsof_accum* acc, int which, int64_t snapshot, int memsize, double v_ds, double* d_out);
/* Frame-driven variant of the same device ODE (simulation/simulationcode_v4_transistor_uav.m:146-227,332-347),
 * float64: imgs = HOST compressed frames [n_frames][H][W] in [0,1]; per frame pair the drive voltage comes
 * from |a-b|*256 through the piecewise map (th1, th2) and modulatefunc, followed by n_sub_steps Euler sub-steps
 * of dt/n_sub_steps.  w_out [H][W]; res_out [n_frames][H][W] (initial array, then one snapshot per pair). */
int nsof_accum_frames_f64(nsof_ctx* ctx, const double* imgs, int n_frames, int height, int width, double dt,
                          int n_sub_steps, double th1, double th2, double* w_out, double* res_out);
/* slice_indices() of event_mem_sim.py:78-83 on a HOST timestamp array: returns the number
 * of bounds and fills idx (if not NULL) with up to cap entries. */
int64_t nsof_accum_slice_bounds(const int64_t* t, int64_t n, int64_t slice_us, int64_t* idx, int64_t cap);

/* ---- ROI gating: the rectangles the gated path crops -------------------------------------------- */
/* opticalFlow3D's gating arithmetic (optical_flow_seg.py:211-252 with :115-121, :426-435) for ONE gating slice, pure
 * host code (maps are at most 24 x 13 cells): current [rows][cols] (device currents in ampere, the constructed3DMatrix
 * slice) -> gray = uint8(clip(-3366/log10(I) - 306, 0, 255)) -> cells with gray >= thres inside the
 * (frame_h / memsize) x (frame_w / memsize) transition picture -> connected components (connectivity 4 or 8, labelled
 * in raster order of their first cell) -> per component (flag 1) or for the union box (flag 2) the rectangle
 * x0 = max(x*memsize - extend_left, 0), y0 = max(y*memsize - extend_upper, 0), x1 = min((x+a)*memsize + extend_right,
 * frame_w), y1 = min((y+b)*memsize + extend_lower, frame_h).  rects receives up to max_rects rows of (x0, y0, x1, y1);
 * returns the number of rectangles (which may exceed max_rects: call again with more room), 0 when nothing crosses
 * the threshold, or a negative nsof_status. */
int nsof_roi_from_surface(const double* current, int rows, int cols, int frame_w, int frame_h, int memsize, int thres,
                          int extend_left, int extend_right, int extend_upper, int extend_lower, int connectivity,
                          int flag, int* rects, int max_rects);

/* Device twin of nsof_roi_from_surface (replaces optical_flow_seg.py:115-121,211-252 for maps that are already in HBM):
 * n_maps gating maps at once, one wavefront each -- d_current: device currents, double, map k at d_current + k*map_stride
 * (rows x cols cells, at most 64 x 64); gray map, threshold, connected components in raster order (bit-parallel flood
 * fill), rectangles as above.  d_counts[k] = number of rectangles of map k (may exceed max_rects; only the first max_rects
 * are stored), d_rects[k][max_rects][4] = (x0, y0, x1, y1); d_gray (optional, may be NULL): the 8-bit gating maps
 * [n_maps][rows][cols].  All DEVICE memory; asynchronous on the context's stream. */
int nsof_roi_from_surface_dev(nsof_ctx* ctx, const double* d_current, int n_maps, size_t map_stride, int rows, int cols,
                              int frame_w, int frame_h, int memsize, int thres, int extend_left, int extend_right,
                              int extend_upper, int extend_lower, int connectivity, int flag, int max_rects, int* d_counts,
                              int* d_rects, unsigned char* d_gray);

/* ---- next: motion-segmentation head on the flow field --------------------- */
/* Replaces, in /root/reference/optical_flow_seg.py, the chain
 *   mag, ang = cv2.cartToPolar(flow_x, flow_y)                         :283, :503
 *   motion_mask[mag > SEG_TH] = 255                                     :346-347
 *   kernel = cv2.getStructuringElement(cv2.MORPH_ELLIPSE, (10, 10))     :350
 *   5 x { cv2.dilate(mask, kernel); cv2.erode(mask, kernel) }           :351-353
 *   cv2.threshold(mask, 1, 255, cv2.THRESH_BINARY)                      :356
 * (process_flow_region :322-357; baseline copy :503-537).  Both operations take
 * dst(x,y) = max|min over element(i,j) != 0 of src(x + j - ax, y + i - ay), anchor = ksize/2,
 * pixels outside the image do not take part (cv2's default morphology border). */
enum { NSOF_MORPH_RECT = 0, NSOF_MORPH_CROSS = 1, NSOF_MORPH_ELLIPSE = 2 };   /* cv2.MORPH_RECT/CROSS/ELLIPSE */
enum { NSOF_MORPH_ERODE = 0, NSOF_MORPH_DILATE = 1 };
/* cv2.getStructuringElement(shape, (kw, kh)) -> out[kh][kw] of 0/1 (HOST). */
int nsof_structuring_element(int shape, int kw, int kh, uint8_t* out);
/* cv2.dilate / cv2.erode for two-valued masks on DEVICE memory: non-zero source pixels count as set, the
 * result is 0/255.  elem = HOST [kh][kw] (non-zero = member), up to 32x32 with at most 8 distinct rows;
 * anchor (-1,-1) = centre; `iterations` repeats the operation.  strides in bytes. */
int nsof_morph_binary_u8_dev(nsof_ctx* ctx, int op, const uint8_t* d_src, ptrdiff_t src_stride, int width,
                             int height, const uint8_t* elem, int kw, int kh, int anchor_x, int anchor_y,
                             int iterations, uint8_t* d_dst, ptrdiff_t dst_stride);
/* The whole head on a DEVICE flow field [h][w][2] float32 (row stride in floats, even): mask = 255 where
 * sqrt(u^2 + v^2) (in double, as cartToPolar sees the float64 canvas) > thresh, then `iterations` x
 * (dilate, erode) with the ksize x ksize ellipse, all in one pass over the flow plus one fused launch on the
 * bit-packed mask.  d_mask [h][mask_stride] uint8. */
int nsof_motion_mask_dev(nsof_ctx* ctx, const float* d_flow, ptrdiff_t flow_stride_floats, int width, int height,
                         double thresh, int ksize, int iterations, uint8_t* d_mask, ptrdiff_t mask_stride);
/* Same with HOST pointers (flow row stride in bytes: ROI views of the flow canvas are passed as they are). */
int nsof_motion_mask(nsof_ctx* ctx, const float* flow, ptrdiff_t flow_stride_bytes, int width, int height,
                     double thresh, int ksize, int iterations, uint8_t* mask, ptrdiff_t mask_stride);

/* ---- next: frame prediction by flow warp + SSIM --------------------------- */
/* Replaces, in /root/reference/optical_flow_prediction.py,
 *   flow_map = (grid + flow).astype(np.float32)                                         :289-290, :338-339, :581-582
 *   cv2.remap(next_frame[:,:,c], flow_map[...,0], flow_map[...,1], cv2.INTER_LINEAR,
 *             borderMode=cv2.BORDER_REPLICATE)      (gated path)                        :293-300, :342-349
 *   cv2.remap(next_frame[:,:,c], map_x, map_y, cv2.INTER_LINEAR)   (baseline, constant 0) :584-586
 *   structural_similarity(true[:,:,2], prediction[:,:,2], data_range=255.0)             :113-115
 * remap follows cv2's 8-bit fixed point: map rounded to 1/32 px (half to even), integer part saturated to int16,
 * 15-bit weights, (sum + 2^14) >> 15. */
enum { NSOF_BORDER_CONSTANT = 0, NSOF_BORDER_REPLICATE = 1 };   /* cv2.BORDER_CONSTANT / cv2.BORDER_REPLICATE */
/* 8-bit luma of an interleaved 3-channel DEVICE frame as cv2.cvtColor computes it (fixed point, (sum + 2^14) >> 15):
 * bgr_weights = 0 -> COLOR_RGB2GRAY weights in memory order (what the scripts apply to imread's B,G,R frames,
 * optical_flow_seg.py:442-443), 1 -> COLOR_BGR2GRAY (:447).  Strides in bytes.  Lets decoded frames stay in HBM up
 * to the flow call. */
int nsof_gray_u8_dev(nsof_ctx* ctx, const uint8_t* d_src, ptrdiff_t src_stride, int width, int height, int bgr_weights,
                     uint8_t* d_dst, ptrdiff_t dst_stride);
/* cv2.remap for uint8 sources with 1 or 3 interleaved channels and two float32 maps, all on the DEVICE.
 * Strides of the images in bytes, of the maps in floats.  Source up to 32767 x 32767. */
int nsof_remap_linear_u8_dev(nsof_ctx* ctx, const uint8_t* d_src, ptrdiff_t src_stride, int src_w, int src_h,
                             int channels, const float* d_map_x, ptrdiff_t map_x_stride_floats,
                             const float* d_map_y, ptrdiff_t map_y_stride_floats, int dst_w, int dst_h,
                             int border_mode, int border_value, uint8_t* d_dst, ptrdiff_t dst_stride);
/* The prediction step for the region [y0,y1) x [x0,x1) of a frame, fused: map = float32(float64(grid) + sign *
 * float64(flow)) is formed in the kernel from the DEVICE flow canvas [height][flow_stride_floats] (u,v) and the
 * region of d_out (a frame-sized image; the caller pre-fills it with the frame, prediction.py:263) is overwritten
 * with the warped d_frame.  sign = -1 for Farneback flow (prediction.py:545). */
int nsof_predict_warp_u8_dev(nsof_ctx* ctx, const uint8_t* d_frame, ptrdiff_t frame_stride, int width, int height,
                             int channels, const float* d_flow, ptrdiff_t flow_stride_floats, int sign,
                             int x0, int y0, int x1, int y1, int border_mode, uint8_t* d_out, ptrdiff_t out_stride);
/* Same with HOST pointers; flow_crop points at the region's first flow vector (row stride in bytes), i.e. the view
 * flow[y0:y1, x0:x1] of a float32 canvas.  Only the region of `out` is written. */
int nsof_predict_warp_u8(nsof_ctx* ctx, const uint8_t* frame, ptrdiff_t frame_stride, int width, int height,
                         int channels, const float* flow_crop, ptrdiff_t flow_stride_bytes, int sign,
                         int x0, int y0, int x1, int y1, int border_mode, uint8_t* out, ptrdiff_t out_stride);
/* skimage.metrics.structural_similarity(a, b, data_range=...) with its defaults (7x7 uniform window, sample
 * covariance, K1 0.01, K2 0.03) for 8-bit images on the DEVICE; pixel_step selects one channel of an interleaved
 * image (3 for true[:,:,2] with the pointer advanced by 2).  *ssim_out is a HOST double; synchronises. */
int nsof_ssim_u8_dev(nsof_ctx* ctx, const uint8_t* d_a, ptrdiff_t a_stride, int a_pixel_step, const uint8_t* d_b,
                     ptrdiff_t b_stride, int b_pixel_step, int width, int height, double data_range,
                     double* ssim_out);
int nsof_ssim_u8(nsof_ctx* ctx, const uint8_t* a, ptrdiff_t a_stride, int a_pixel_step, const uint8_t* b,
                 ptrdiff_t b_stride, int b_pixel_step, int width, int height, double data_range, double* ssim_out);

#ifdef __cplusplus
}
#endif
#endif /* NSOF_H */
