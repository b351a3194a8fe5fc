// libnsof.so host side: context, error channel, profiling hooks, Farneback level driver.
// The level loop mirrors the driver of the reference's flow backend
// (cv2.calcOpticalFlowFarneback, called at /root/reference/optical_flow_seg.py:203):
// coarsest level first, every level resampled from the blurred FULL-RES frame.
#include <cctype>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <algorithm>
#include <cstring>

#include <sys/syscall.h>
#include <unistd.h>

#include "nsof_internal.h"

static char g_create_err[512] = "";

int nsof_set_error(nsof_ctx* ctx, int code, const char* fmt, ...)
{
    char* dst = ctx ? ctx->err : g_create_err;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(dst, 512, fmt, ap);
    va_end(ap);
    return code;
}

int nsof_ws_reserve(nsof_ctx* ctx, void** buf, size_t* cur, size_t need)
{
    if (*cur >= need) return NSOF_OK;
    if (*buf) {
        NSOF_HIP(ctx, hipStreamSynchronize(ctx->stream));
        NSOF_HIP(ctx, hipFree(*buf));
        *buf = nullptr;
        *cur = 0;
    }
    hipError_t e = hipMalloc(buf, need);
    if (e != hipSuccess) {
        *buf = nullptr;
        return nsof_set_error(ctx, NSOF_ENOMEM, "hipMalloc(%zu) failed: %s", need, hipGetErrorString(e));
    }
    *cur = need;
    return NSOF_OK;
}

int nsof_xsync_reserve(nsof_ctx* ctx, size_t carry_bytes, unsigned long long** carry, unsigned** tickets, unsigned** err)
{
    if (!ctx->x_sync) {
        NSOF_HIP(ctx, hipMalloc((void**)&ctx->x_sync, 2048));
        NSOF_HIP(ctx, hipMemsetAsync(ctx->x_sync, 0, 2048, ctx->stream));
    }
    if (carry_bytes > ctx->x_carry_bytes) {
        carry_bytes = (carry_bytes + (carry_bytes >> 3) + 4095) & ~(size_t)4095;   // some slack: shapes vary from call to call
        if (int rc = nsof_ws_reserve(ctx, (void**)&ctx->x_carry, &ctx->x_carry_bytes, carry_bytes)) return rc;
        NSOF_HIP(ctx, hipMemsetAsync(ctx->x_carry, 0, ctx->x_carry_bytes, ctx->stream));   // tag 0 = never written
    }
    *carry = ctx->x_carry;
    *tickets = ctx->x_sync;
    *err = ctx->x_sync + 256;
    ctx->x_dirty = true;
    return NSOF_OK;
}

int nsof_xsync_check(nsof_ctx* ctx)
{
    if (!ctx->x_dirty || !ctx->x_sync) return NSOF_OK;
    ctx->x_dirty = false;
    unsigned w = 0;
    NSOF_HIP(ctx, hipMemcpy(&w, ctx->x_sync + 256, sizeof(w), hipMemcpyDeviceToHost));
    if (w) {
        NSOF_HIP(ctx, hipMemset(ctx->x_sync + 256, 0, sizeof(w)));
        return nsof_set_error(ctx, NSOF_EDEVICE, "exact-order iteration: a hand-over between workgroups / waves never arrived (flags %u); results are invalid", w);
    }
    return NSOF_OK;
}

int nsof_stream_sync_checked(nsof_ctx* ctx)
{
    NSOF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return nsof_xsync_check(ctx);
}

// ---- profiling ---------------------------------------------------------------------------
nsof_prof_scope::nsof_prof_scope(nsof_ctx* c, int k) : ctx(c), id(k), on((c->prof_mask >> k) & 1u)
{
    if (!on) return;
    nsof_prof_slot& s = ctx->prof[id];
    if (s.used == s.start.size()) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { on = false; return; }
        s.start.push_back(a);
        s.stop.push_back(b);
    }
    hipEventRecord(s.start[s.used], ctx->stream);
}
nsof_prof_scope::~nsof_prof_scope()
{
    if (!on) return;
    nsof_prof_slot& s = ctx->prof[id];
    hipEventRecord(s.stop[s.used], ctx->stream);
    s.used++;
}

extern "C" int nsof_prof_enable(nsof_ctx* ctx, unsigned mask)
{
    if (!ctx) return NSOF_EINVAL;
    ctx->prof_mask = mask & ((1u << NSOF_K_COUNT) - 1);
    return NSOF_OK;
}

extern "C" int nsof_prof_collect(nsof_ctx* ctx, int id, double* total_ms, long long* launches)
{
    if (!ctx || id < 0 || id >= NSOF_K_COUNT) return NSOF_EINVAL;
    NSOF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->side) NSOF_HIP(ctx, hipStreamSynchronize(ctx->side));
    nsof_prof_slot& s = ctx->prof[id];
    for (size_t i = 0; i < s.used; i++) {
        float ms = 0;
        NSOF_HIP(ctx, hipEventElapsedTime(&ms, s.start[i], s.stop[i]));
        s.acc_ms += ms;
        s.acc_launches++;
    }
    s.used = 0;
    if (total_ms) *total_ms = s.acc_ms;
    if (launches) *launches = s.acc_launches;
    s.acc_ms = 0;
    s.acc_launches = 0;
    return NSOF_OK;
}

// ---- page-locked host memory next to the GPU ------------------------------------------------------------------
// On a two-socket host a pinned buffer on the far socket halves the PCIe copy rate (measured on the MI355X boxes:
// 28 instead of 57 GB/s).  The GPU's NUMA node comes from sysfs (PCI bus id -> numa_node); the allocation runs under
// a temporary MPOL_PREFERRED policy for that node (hipHostMallocNumaUser makes the runtime honour it).  Every step
// is best effort: without sysfs / the syscall the default placement is used.
int nsof_gpu_numa_node(int device)
{
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    for (char* c = bus; *c; c++) *c = (char)tolower(*c);
    char path[160];
    snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bus);
    FILE* f = fopen(path, "r");
    if (!f) return -1;
    int node = -1;
    if (fscanf(f, "%d", &node) != 1) node = -1;
    fclose(f);
    return node;
}

void* nsof_pinned_alloc(int device, size_t bytes)
{
    void* p = nullptr;
    const int node = nsof_gpu_numa_node(device);
    bool policy = false;
    // the calling thread's own policy (an application may run under numactl --membind / --interleave): saved and restored
    int old_mode = 0;
    unsigned long old_mask[16] = {0};
    const bool have_old = syscall(SYS_get_mempolicy, &old_mode, old_mask, 8 * sizeof(old_mask) + 1, nullptr, 0) == 0;
    if (node >= 0 && node < 1024 && have_old) {
        unsigned long mask[16] = {0};
        mask[node / (8 * sizeof(unsigned long))] = 1ul << (node % (8 * sizeof(unsigned long)));
        policy = syscall(SYS_set_mempolicy, 1 /* MPOL_PREFERRED */, mask, 8 * sizeof(mask) + 1) == 0;
    }
    hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, policy ? hipHostMallocNumaUser : hipHostMallocDefault);
    if (policy) (void)syscall(SYS_set_mempolicy, old_mode, old_mode == 0 ? nullptr : old_mask, old_mode == 0 ? 0 : 8 * sizeof(old_mask) + 1);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}

int nsof_hstage_reserve(nsof_ctx* ctx, size_t need)
{
    if (ctx->hstage_bytes >= need) return NSOF_OK;
    NSOF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->hstage) hipHostFree(ctx->hstage);
    ctx->hstage = nullptr;
    ctx->hstage_bytes = 0;
    ctx->hstage = nsof_pinned_alloc(ctx->device, need);
    if (!ctx->hstage) return nsof_set_error(ctx, NSOF_ENOMEM, "hipHostMalloc(%zu) failed", need);
    ctx->hstage_bytes = need;
    return NSOF_OK;
}

extern "C" const char* nsof_kernel_name(int id)
{
    static const char* names[NSOF_K_COUNT] = {"prep", "polyexp", "flow_upsample", "update_matrices", "blur_solve",
                                              "accum_update", "iterate", "mask_pack", "morph_chain", "remap", "ssim"};
    return (id >= 0 && id < NSOF_K_COUNT) ? names[id] : "?";
}

// ---- context -----------------------------------------------------------------------------
extern "C" int nsof_abi_version(void) { return NSOF_ABI_VERSION; }

extern "C" int nsof_create(int device, nsof_ctx** out)
{
    if (!out) return NSOF_EINVAL;
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return nsof_set_error(nullptr, NSOF_EDEVICE, "no HIP device available (%s); libnsof has no CPU fallback",
                              e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device < 0 || device >= count)
        return nsof_set_error(nullptr, NSOF_EINVAL, "device %d out of range (0..%d)", device, count - 1);
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess)
        return nsof_set_error(nullptr, NSOF_EDEVICE, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return nsof_set_error(nullptr, NSOF_EDEVICE, "device %d is %s; libnsof is built for gfx950 only", device,
                              prop.gcnArchName);
    if ((e = hipSetDevice(device)) != hipSuccess)
        return nsof_set_error(nullptr, NSOF_EDEVICE, "hipSetDevice: %s", hipGetErrorString(e));
    nsof_ctx* ctx = new (std::nothrow) nsof_ctx();
    if (!ctx) return nsof_set_error(nullptr, NSOF_ENOMEM, "out of host memory");
    ctx->device = device;
    if ((e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking)) != hipSuccess) {
        delete ctx;
        return nsof_set_error(nullptr, NSOF_EDEVICE, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    ctx->stream = ctx->own_stream;
    if (const char* e = getenv("NSOF_POLYEXP_F32")) ctx->opt_polyexp_f32 = (e[0] && e[0] != '0') ? 1 : 0;
    if (const char* e = getenv("NSOF_EXACT_ROWSUMS")) ctx->opt_exact_rowsums = (e[0] && e[0] != '0') ? 1 : 0;
    if (const char* e = getenv("NSOF_PYR_FMA")) ctx->opt_pyr_fma = (e[0] && e[0] != '0') ? 1 : 0;
    if (const char* e = getenv("NSOF_LAT_JOBS")) ctx->opt_small_batch_jobs = atoi(e) < 0 ? 0 : atoi(e);
    if (const char* e = getenv("NSOF_ROW_BANDS")) {
        const int v = atoi(e);
        ctx->opt_row_bands = v < 0 || v == 2 || v == 3 ? 0 : v;
    }
    *out = ctx;
    return NSOF_OK;
}

extern "C" void nsof_destroy(nsof_ctx* ctx)
{
    if (!ctx) return;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    for (auto& s : ctx->prof) {
        for (auto ev : s.start) hipEventDestroy(ev);
        for (auto ev : s.stop) hipEventDestroy(ev);
    }
    if (ctx->ws) hipFree(ctx->ws);
    if (ctx->stage) hipFree(ctx->stage);
    if (ctx->tmp) hipFree(ctx->tmp);
    if (ctx->hstage) hipHostFree(ctx->hstage);
    if (ctx->het_h) hipHostFree(ctx->het_h);
    if (ctx->het_d) hipFree(ctx->het_d);
    if (ctx->x_carry) hipFree(ctx->x_carry);
    if (ctx->roi_tmp) hipFree(ctx->roi_tmp);
    if (ctx->paste_h) hipHostFree(ctx->paste_h);
    if (ctx->paste_d) hipFree(ctx->paste_d);
    if (ctx->paste_ev) hipEventDestroy(ctx->paste_ev);
    if (ctx->x_sync) hipFree(ctx->x_sync);
    for (auto ev : ctx->het_ev)
        if (ev) hipEventDestroy(ev);
    nsof_pipe_destroy(ctx);
    for (auto ev : ctx->ov_events) hipEventDestroy(ev);
    if (ctx->side) hipStreamDestroy(ctx->side);
    if (ctx->own_stream) hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

extern "C" const char* nsof_last_error(const nsof_ctx* ctx) { return ctx ? ctx->err : g_create_err; }

extern "C" int nsof_set_stream(nsof_ctx* ctx, void* s)
{
    if (!ctx) return NSOF_EINVAL;
    NSOF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = s ? (hipStream_t)s : ctx->own_stream;
    return NSOF_OK;
}

extern "C" int nsof_set_option(nsof_ctx* ctx, int option, int value)
{
    if (!ctx) return NSOF_EINVAL;
    if (option == NSOF_OPT_POLYEXP_F32 || option == NSOF_OPT_EXACT_ROWSUMS || option == NSOF_OPT_PYR_FMA) {
        if (value != 0 && value != 1) return nsof_set_error(ctx, NSOF_EINVAL, "option %d takes 0 or 1", option);
        (option == NSOF_OPT_POLYEXP_F32 ? ctx->opt_polyexp_f32 : option == NSOF_OPT_PYR_FMA ? ctx->opt_pyr_fma : ctx->opt_exact_rowsums) = value;
        return NSOF_OK;
    }
    if (option == NSOF_OPT_ROW_BANDS) {
        if (value < 0 || value == 2 || value == 3)
            return nsof_set_error(ctx, NSOF_EINVAL, "NSOF_OPT_ROW_BANDS: 0 (off), 1 (automatic) or a row count >= 4");
        ctx->opt_row_bands = value;
        return NSOF_OK;
    }
    if (option == NSOF_OPT_SMALL_BATCH_JOBS) {
        if (value < 0) return nsof_set_error(ctx, NSOF_EINVAL, "NSOF_OPT_SMALL_BATCH_JOBS: a job count >= 0");
        ctx->opt_small_batch_jobs = value;
        return NSOF_OK;
    }
    if (option == NSOF_OPT_DEBUG_FAULT) {
        if (value < 0 || value > 3) return nsof_set_error(ctx, NSOF_EINVAL, "NSOF_OPT_DEBUG_FAULT: bits 0..1");
        ctx->dbg_fault = value;
        return NSOF_OK;
    }
    return nsof_set_error(ctx, NSOF_EINVAL, "unknown option %d", option);
}

extern "C" int nsof_get_option(const nsof_ctx* ctx, int option, int* value)
{
    if (!ctx || !value) return NSOF_EINVAL;
    if (option == NSOF_OPT_POLYEXP_F32) {
        *value = ctx->opt_polyexp_f32;
        return NSOF_OK;
    }
    if (option == NSOF_OPT_EXACT_ROWSUMS) {
        *value = ctx->opt_exact_rowsums;
        return NSOF_OK;
    }
    if (option == NSOF_OPT_ROW_BANDS) {
        *value = ctx->opt_row_bands;
        return NSOF_OK;
    }
    if (option == NSOF_OPT_PYR_FMA) {
        *value = ctx->opt_pyr_fma;
        return NSOF_OK;
    }
    if (option == NSOF_OPT_SMALL_BATCH_JOBS) {
        *value = ctx->opt_small_batch_jobs;
        return NSOF_OK;
    }
    if (option == NSOF_OPT_DEBUG_FAULT) {
        *value = ctx->dbg_fault;
        return NSOF_OK;
    }
    return NSOF_EINVAL;
}

extern "C" int nsof_synchronize(nsof_ctx* ctx)
{
    if (!ctx) return NSOF_EINVAL;
    return nsof_stream_sync_checked(ctx);
}

// ---- filter taps (host, double precision as the reference library computes them) -----------
static int round_half_even(double v) { return (int)lrint(v); }

int nsof_host_blur_taps(int ksize, double sigma, nsof_blur_taps* out)
{
    if (ksize < 1 || (ksize & 1) == 0 || ksize > NSOF_MAX_BLUR_TAPS - 1) return NSOF_EUNSUPPORTED;
    out->ksize = ksize;
    memset(out->k, 0, sizeof(out->k));
    if (sigma <= 0 && ksize <= 9) {  // fixed small tables
        static const double t1[] = {1.};
        static const double t3[] = {0.25, 0.5, 0.25};
        static const double t5[] = {0.0625, 0.25, 0.375, 0.25, 0.0625};
        static const double t7[] = {0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125};
        static const double t9[] = {4. / 256, 13. / 256, 30. / 256, 51. / 256, 60. / 256, 51. / 256, 30. / 256, 13. / 256, 4. / 256};
        const double* t = ksize == 1 ? t1 : ksize == 3 ? t3 : ksize == 5 ? t5 : ksize == 7 ? t7 : t9;
        for (int i = 0; i < ksize; i++) out->k[i] = (float)t[i];
        return NSOF_OK;
    }
    const double sg = sigma > 0 ? sigma : ((ksize - 1) * 0.5 - 1) * 0.3 + 0.8;
    const double scale2 = -0.125 / (sg * sg);
    const int h = (ksize - 1) / 2;
    double v[NSOF_MAX_BLUR_TAPS], sum = 0;
    for (int i = 0, x = 1 - ksize; i < h; i++, x += 2) {
        v[i] = std::exp((double)(x * x) * scale2);
        sum += v[i];
    }
    sum = sum * 2.0 + 1.0;
    const double inv = 1.0 / sum;
    for (int i = 0; i < h; i++) out->k[i] = out->k[ksize - 1 - i] = (float)(v[i] * inv);
    out->k[h] = (float)inv;
    return NSOF_OK;
}

static bool chol_inv6(const double A[6][6], double inv[6][6])
{
    double L[6][6] = {{0}};
    for (int i = 0; i < 6; i++)
        for (int j = 0; j <= i; j++) {
            double s = A[i][j];
            for (int k = 0; k < j; k++) s -= L[i][k] * L[j][k];
            if (i == j) {
                if (s <= 0) return false;
                L[i][i] = std::sqrt(s);
            } else
                L[i][j] = s / L[j][j];
        }
    for (int c = 0; c < 6; c++) {
        double y[6], x[6];
        for (int i = 0; i < 6; i++) {
            double s = (i == c) ? 1. : 0.;
            for (int k = 0; k < i; k++) s -= L[i][k] * y[k];
            y[i] = s / L[i][i];
        }
        for (int i = 5; i >= 0; i--) {
            double s = y[i];
            for (int k = i + 1; k < 6; k++) s -= L[k][i] * x[k];
            x[i] = s / L[i][i];
        }
        for (int i = 0; i < 6; i++) inv[i][c] = x[i];
    }
    return true;
}

int nsof_host_poly_taps(int n, double sigma, nsof_poly_taps* out)
{
    if (n < 1 || n > NSOF_MAX_POLY_N) return NSOF_EUNSUPPORTED;
    if (sigma < FLT_EPSILON) sigma = n * 0.3;
    float g[2 * NSOF_MAX_POLY_N + 1], xg[2 * NSOF_MAX_POLY_N + 1], xxg[2 * NSOF_MAX_POLY_N + 1];
    double s = 0.;
    for (int x = -n; x <= n; x++) {
        g[x + n] = (float)std::exp(-x * x / (2 * sigma * sigma));
        s += g[x + n];
    }
    s = 1. / s;
    for (int x = -n; x <= n; x++) {
        g[x + n] = (float)(g[x + n] * s);
        xg[x + n] = (float)(x * g[x + n]);
        xxg[x + n] = (float)(x * x * g[x + n]);
    }
    double G[6][6] = {{0}}, inv[6][6];
    for (int y = -n; y <= n; y++)
        for (int x = -n; x <= n; x++) {
            G[0][0] += g[y + n] * g[x + n];
            G[1][1] += g[y + n] * g[x + n] * x * x;
            G[3][3] += g[y + n] * g[x + n] * x * x * x * x;
            G[5][5] += g[y + n] * g[x + n] * x * x * y * y;
        }
    G[2][2] = G[0][3] = G[0][4] = G[3][0] = G[4][0] = G[1][1];
    G[4][4] = G[3][3];
    G[3][4] = G[4][3] = G[5][5];
    if (!chol_inv6(G, inv)) return NSOF_EINVAL;
    memset(out, 0, sizeof(*out));
    out->n = n;
    for (int k = 0; k <= n; k++) {
        out->g[k] = g[n + k];
        out->xg[k] = xg[n + k];
        out->xxg[k] = xxg[n + k];
        out->dg[k] = (double)g[n + k];
        out->dxxg[k] = (double)xxg[n + k];
    }
    out->ig11 = inv[1][1];
    out->ig03 = inv[0][3];
    out->ig33 = inv[3][3];
    out->ig55 = inv[5][5];
    return NSOF_OK;
}

// ---- level geometry ------------------------------------------------------------------------
extern "C" int nsof_farneback_effective_levels(int width, int height, double pyr_scale, int levels)
{
    int k;
    double scale = 1;
    for (k = 0; k < levels; k++) {
        scale *= pyr_scale;
        if (width * scale < 32 || height * scale < 32) break;
    }
    return k;
}

extern "C" int nsof_farneback_level_size(int width, int height, double pyr_scale, int level, int* lw, int* lh,
                                          int* ksize, double* sigma)
{
    if (width < 1 || height < 1 || level < 0 || !(pyr_scale > 0) || !(pyr_scale < 1)) return NSOF_EINVAL;
    double scale = 1;
    for (int i = 0; i < level; i++) scale *= pyr_scale;
    const double sg = (1. / scale - 1) * 0.5;
    int sz = round_half_even(sg * 5) | 1;
    if (sz < 3) sz = 3;
    if (lw) *lw = round_half_even(width * scale);
    if (lh) *lh = round_half_even(height * scale);
    if (ksize) *ksize = sz;
    if (sigma) *sigma = sg;
    return NSOF_OK;
}

int nsof_check_farneback_params(nsof_ctx* ctx, int width, int height, double pyr_scale, int levels, int winsize,
                                int iterations, int poly_n, int flags)
{
    if (width < 1 || height < 1) return nsof_set_error(ctx, NSOF_ESHAPE, "empty image %dx%d", width, height);
    if ((long long)width * height > (1ll << 27))   // kernels address one image with 32-bit byte offsets
        return nsof_set_error(ctx, NSOF_EUNSUPPORTED, "image %dx%d exceeds 2^27 pixels", width, height);
    if (!(pyr_scale > 0) || !(pyr_scale < 1))
        return nsof_set_error(ctx, NSOF_EINVAL, "pyr_scale=%g must be in (0,1)", pyr_scale);
    if (winsize == 1)  // upstream's running sums are ill-formed for a 1x1 window (m = 0); never used by the reference
        return nsof_set_error(ctx, NSOF_EUNSUPPORTED, "winsize=1 is not supported");
    if (levels < 0 || winsize < 1 || iterations < 0)
        return nsof_set_error(ctx, NSOF_EINVAL, "levels=%d winsize=%d iterations=%d invalid", levels, winsize,
                              iterations);
    if (poly_n < 1 || poly_n > NSOF_MAX_POLY_N)
        return nsof_set_error(ctx, poly_n < 1 ? NSOF_EINVAL : NSOF_EUNSUPPORTED, "poly_n=%d outside 1..%d", poly_n,
                              NSOF_MAX_POLY_N);
    if (flags != 0)
        return nsof_set_error(ctx, NSOF_EUNSUPPORTED,
                              "flags=%d: OPTFLOW_USE_INITIAL_FLOW / OPTFLOW_FARNEBACK_GAUSSIAN are not implemented "
                              "(the reference always passes flags=0)", flags);
    return NSOF_OK;
}

// ---- stage entry points ---------------------------------------------------------------------
extern "C" int nsof_stage_pyr_level(nsof_ctx* ctx, int n_img, const uint8_t* d_src, ptrdiff_t row_stride,
                                    ptrdiff_t img_stride, int width, int height, double pyr_scale, int level,
                                    float* d_out)
{
    if (!ctx || !d_src || !d_out || n_img < 1) return NSOF_EINVAL;
    int wk, hk, ks;
    double sg;
    int rc = nsof_farneback_level_size(width, height, pyr_scale, level, &wk, &hk, &ks, &sg);
    if (rc) return nsof_set_error(ctx, rc, "bad level geometry");
    nsof_blur_taps taps;
    if ((rc = nsof_host_blur_taps(ks, sg, &taps)))
        return nsof_set_error(ctx, rc, "pyramid blur kernel size %d unsupported (max %d)", ks, NSOF_MAX_BLUR_TAPS - 1);
    return NSOF_PYR_SEL(ctx, nsof_launch_prep, n_img, d_src, row_stride, img_stride, width, height, wk, hk, taps, d_out);
}

__global__ void k_recip_probe(long long n, const double* __restrict__ x, double* __restrict__ fast, double* __restrict__ ieee)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        fast[i] = nsof_recip_normal(x[i]);
        ieee[i] = 1. / x[i];
    }
}

extern "C" int nsof_stage_recip(nsof_ctx* ctx, long long n, const double* d_x, double* d_out, double* d_ieee)
{
    if (!ctx || !d_x || !d_out || !d_ieee || n < 1) return NSOF_EINVAL;
    hipLaunchKernelGGL(k_recip_probe, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, n, d_x, d_out, d_ieee);
    NSOF_HIP(ctx, hipGetLastError());
    return NSOF_OK;
}

extern "C" int nsof_stage_polyexp(nsof_ctx* ctx, int n_img, const float* d_img, int width, int height, int poly_n,
                                  double poly_sigma, float* d_R)
{
    if (!ctx || !d_img || !d_R || n_img < 1 || width < 1 || height < 1) return NSOF_EINVAL;
    nsof_poly_taps taps;
    int rc = nsof_host_poly_taps(poly_n, poly_sigma, &taps);
    if (rc) return nsof_set_error(ctx, rc, "poly_n=%d unsupported", poly_n);
    return nsof_launch_polyexp(ctx, n_img, d_img, width, height, taps, d_R);
}

extern "C" int nsof_stage_update_matrices(nsof_ctx* ctx, int n_pairs, const float* d_R, const float* d_flow,
                                          int width, int height, float* d_M)
{
    if (!ctx || !d_R || !d_flow || !d_M || n_pairs < 1 || width < 1 || height < 1) return NSOF_EINVAL;
    const size_t plane = (size_t)width * height;
    return nsof_launch_update_matrices(ctx, n_pairs, d_R, d_R + 5 * plane, 10 * plane, d_flow, width, height, d_M);
}

extern "C" int nsof_stage_blur_solve(nsof_ctx* ctx, int n_pairs, const float* d_M, int width, int height,
                                     int winsize, float* d_flow)
{
    if (!ctx || !d_M || !d_flow || n_pairs < 1 || width < 1 || height < 1 || winsize < 2) return NSOF_EINVAL;
    return nsof_launch_blur_solve(ctx, n_pairs, d_M, width, height, winsize, d_flow);
}

extern "C" int nsof_stage_iterate(nsof_ctx* ctx, int n_pairs, const float* d_R, const float* d_flow_in, int width,
                                  int height, int winsize, float* d_flow_out)
{
    if (!ctx || !d_R || !d_flow_in || !d_flow_out || d_flow_in == d_flow_out || n_pairs < 1 || width < 1 || height < 1)
        return NSOF_EINVAL;
    if (!nsof_iterate_supported(winsize, width, height)) return nsof_set_error(ctx, NSOF_EUNSUPPORTED, "winsize %d not fused", winsize);
    const size_t plane = (size_t)width * height;
    if (ctx->opt_exact_rowsums && nsof_iterate_x_supported(winsize, width, height))
        return nsof_launch_iterate_x(ctx, n_pairs, d_R, d_R + 5 * plane, 10 * plane, d_flow_in, d_flow_out, width, height,
                                     winsize);
    return nsof_launch_iterate(ctx, n_pairs, d_R, d_R + 5 * plane, 10 * plane, d_flow_in, d_flow_out, width, height,
                               winsize);
}

extern "C" int nsof_stage_iterate_upsample(nsof_ctx* ctx, int n_pairs, const float* d_R, const float* d_coarse_flow,
                                           int src_w, int src_h, int width, int height, int winsize, double pyr_scale,
                                           float* d_flow_out)
{
    if (!ctx || !d_R || !d_coarse_flow || !d_flow_out || n_pairs < 1 || width < 1 || height < 1 || src_w < 1 || src_h < 1)
        return NSOF_EINVAL;
    const size_t plane = (size_t)width * height;
    return nsof_launch_iterate_upsample(ctx, n_pairs, d_R, d_R + 5 * plane, 10 * plane, d_coarse_flow, src_w, src_h,
                                        (float)(1. / pyr_scale), d_flow_out, width, height, winsize);
}

extern "C" int nsof_stage_flow_upsample(nsof_ctx* ctx, int n_pairs, const float* d_src, int sw, int sh, float* d_dst,
                                        int dw, int dh, double pyr_scale)
{
    if (!ctx || !d_src || !d_dst || n_pairs < 1 || sw < 1 || sh < 1 || dw < 1 || dh < 1) return NSOF_EINVAL;
    return NSOF_PYR_SEL(ctx, nsof_launch_flow_upsample, n_pairs, d_src, sw, sh, d_dst, dw, dh, (float)(1. / pyr_scale));
}

// ---- the Farneback driver --------------------------------------------------------------------
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Core of both device entry points.  sequence == false: n_pairs independent pairs (d_prev[i], d_next[i]);
// sequence == true: n_pairs + 1 consecutive frames in d_prev (d_next unused), pair i = (frame i, frame i+1) -- every
// frame's pyramid level and polynomial expansion is then computed once and shared by the two pairs it belongs to.
int nsof_farneback_core(nsof_ctx* ctx, bool sequence, int n_pairs, const uint8_t* d_prev, const uint8_t* d_next,
                          ptrdiff_t row_stride, ptrdiff_t pair_stride, int width, int height, float* d_flow,
                          double pyr_scale, int levels, int winsize, int iterations, int poly_n, double poly_sigma,
                          int flags)
{
    if (!ctx) return NSOF_EINVAL;
    if (sequence) d_next = d_prev;
    if (!d_prev || !d_next || !d_flow || n_pairs < 1) return nsof_set_error(ctx, NSOF_EINVAL, "null buffer or n_pairs<1");
    int rc = nsof_check_farneback_params(ctx, width, height, pyr_scale, levels, winsize, iterations, poly_n, flags);
    if (rc) return rc;
    if (row_stride < width) return nsof_set_error(ctx, NSOF_EINVAL, "row_stride < width");
    NSOF_HIP(ctx, hipSetDevice(ctx->device));
    const bool exact = ctx->opt_exact_rowsums != 0;
    // exact row-sum order: ONE fused kernel (k_iterate_x) where the window fits; NSOF_EXACT_IMPL=2k selects the older
    // two-kernel form (column sums through HBM) for A/B runs
    static const bool exact_2k = [] { const char* e = NSOF_AB_GETENV("NSOF_EXACT_IMPL"); return e && e[0] == '2'; }();
    static const char* fused_env0 = NSOF_AB_GETENV("NSOF_FUSED");
    const bool exact_x = exact && !exact_2k && !(fused_env0 && fused_env0[0] == '0') && iterations > 0 &&
                         nsof_iterate_x_supported(winsize, width, height);
    // a batch too small to fill the chip with (strip, image) jobs takes the three-kernel small-batch form of the same
    // order (farneback_iterate_lat.hip; NSOF_OPT_SMALL_BATCH_JOBS)
    const bool exact_lat = exact_x && (long long)n_pairs * ((width + 191) / 192) <= ctx->opt_small_batch_jobs &&
                           (unsigned long long)width * height * 40ull < (1ull << 32);   // its kernels address a pair with 32-bit byte offsets
    static const int exact_chunk = [] {
        const char* e = NSOF_AB_GETENV("NSOF_EXACT_CHUNK");
        const int v = e ? atoi(e) : 64;
        return v < 1 ? 1 : v;
    }();
    if (exact && !exact_x && (sequence || n_pairs > exact_chunk)) {
        // the exact order keeps 40 B/px of column sums (+ 20 B/px of matrices) in HBM: 64 pairs of 1920x1080 at a time
        // (8 GB) fill the GPU -- the row walk has one thread per image row; a sequence is run as its pairs
        const uint8_t* nx = sequence ? d_prev + pair_stride : d_next;
        for (int i = 0; i < n_pairs; i += exact_chunk) {
            const int nb = n_pairs - i < exact_chunk ? n_pairs - i : exact_chunk;
            rc = nsof_farneback_core(ctx, false, nb, d_prev + (ptrdiff_t)i * pair_stride, nx + (ptrdiff_t)i * pair_stride,
                                     row_stride, pair_stride, width, height, d_flow + (size_t)i * width * height * 2,
                                     pyr_scale, levels, winsize, iterations, poly_n, poly_sigma, flags);
            if (rc) return rc;
        }
        return NSOF_OK;
    }

    {
        // A batch whose workspace (56 B/px/pair = 116 MB per 1080p pair: level images, expansions, second flow buffer;
        // + 20 / 40 B/px of matrices / column sums in the unfused / exact forms) would not fit the device's free memory is run in chunks of as many pairs as do
        // fit -- same kernels on sub-ranges of the same buffers, so the result does not depend on the chunking.
        // NSOF_MAX_PAIRS caps the chunk by hand (tests).
        // (the latency schedule of small batches keeps the level images and expansions of EVERY level: at most levels + 1 times those terms)
        const size_t per_pair = (size_t)width * height * ((4 * 2 + 20 * 2) * (exact_lat ? (size_t)levels + 1 : 1) + 8 + 20 +
                                                          (exact && !exact_x ? 40 : 0) + (exact_lat ? 60 : 0)) + 4096;
        size_t fit = ctx->ws_bytes / per_pair;   // what the workspace already holds needs no query (lone calls stay cheap)
        if ((size_t)n_pairs > fit) {
            size_t free_b = 0, total_b = 0;
            NSOF_HIP(ctx, hipMemGetInfo(&free_b, &total_b));
            fit = (size_t)((double)(free_b + ctx->ws_bytes) * 0.92) / per_pair;
        }
        if (const char* e = getenv("NSOF_MAX_PAIRS")) {
            const long v = atol(e);
            if (v >= 1 && (size_t)v < fit) fit = (size_t)v;
        }
        if (fit < 1) fit = 1;
        if (fit > 32767) fit = 32767;   // 2 * pairs images go on gridDim.z of one launch
        if ((size_t)n_pairs > fit) {
            for (int i = 0; i < n_pairs; i += (int)fit) {
                const int nb = n_pairs - i < (int)fit ? n_pairs - i : (int)fit;
                rc = nsof_farneback_core(ctx, sequence, nb, d_prev + (ptrdiff_t)i * pair_stride,
                                         d_next + (ptrdiff_t)i * pair_stride, row_stride, pair_stride, width, height,
                                         d_flow + (size_t)i * width * height * 2, pyr_scale, levels, winsize, iterations,
                                         poly_n, poly_sigma, flags);
                if (rc) return rc;
            }
            return NSOF_OK;
        }
    }

    nsof_poly_taps ptaps;
    if ((rc = nsof_host_poly_taps(poly_n, poly_sigma, &ptaps))) return nsof_set_error(ctx, rc, "poly taps");
    const int L = nsof_farneback_effective_levels(width, height, pyr_scale, levels);

    // workspace: I [n_img][n0] f32, R [n_img][5*n0] f32, S = second flow buffer [B][n0][2]
    // (+ M [B][5][n0] only when the window is too large for the fused iteration kernel)
    const size_t n0 = (size_t)width * height, B = (size_t)n_pairs;
    const size_t n_img = sequence ? B + 1 : 2 * B;   // frames of a sequence, or B prev + B next frames
    // Fused or unfused is decided once for the whole pyramid (inputs below 2x2 take the unfused pair).
    // NSOF_FUSED=0 forces the unfused pair (A/B runs; measured slower even for a lone 1080p pair: 6.1 vs 4.0 ms).
    static const char* fused_env = NSOF_AB_GETENV("NSOF_FUSED");
    // exact row-sum order: the fused two-kernel form (phase A + row scan) where the window fits, else the unfused kernels
    const bool exact_fused = exact && (exact_x || (nsof_iterate_exact_supported(winsize, width, height) && !(fused_env && fused_env[0] == '0')));
    const bool fused = nsof_iterate_supported(winsize, width, height) && !(fused_env && fused_env[0] == '0') &&
                       (!exact || exact_fused);
    const size_t szI = align_up(n_img * n0 * 4, 256), szR = align_up(n_img * 5 * n0 * 4, 256);
    const size_t szS = align_up(B * n0 * 8, 256), szM = fused && !exact_lat ? 0 : align_up(B * 5 * n0 * 4, 256);
    const size_t szV = (exact && !exact_x) || exact_lat ? align_up(B * 5 * n0 * 8, 256) : 0;   // column sums of the two- / three-kernel exact order
    if ((rc = nsof_ws_reserve(ctx, &ctx->ws, &ctx->ws_bytes, szI + szR + szS + szM + szV))) return rc;
    char* base = (char*)ctx->ws;   // (re-derived below if the level overlap grows the workspace)
    float* dI = (float*)base;
    float* dR = (float*)(base + szI);
    float* dS = (float*)(base + szI + szR);
    float* dM = (float*)(base + szI + szR + szS);
    double* dV = (double*)(base + szI + szR + szS + szM);
    // Two flow buffers, A = the caller's output and S = scratch; every level uses their leading B*nk pixels.
    // Each upsample and each fused iteration moves the flow to the other buffer, so the buffer the coarsest
    // level starts in is chosen such that the last iteration of level 0 writes A.
    float* fb[2] = {d_flow, dS};
    // The coarse-to-fine resample is folded into the first iteration of each level when the fused kernel can do
    // it; then only iterations move the flow between the buffers.
    // Measured on MI355X (1080p x 128 pairs): folding costs more in the producers (4 gathers + the resample per
    // row, 168 VGPRs) than the standalone resample kernel saves (24.6 -> 25.5 ms per step), so it is opt-in.
    static const bool fold_env = NSOF_AB_GETENV("NSOF_FOLD_UPSAMPLE") != nullptr;
    const bool fold_ups = fold_env && fused && !exact && iterations > 0 &&
                          nsof_iterate_upsample_supported(winsize, width, height);
    const int flips = fused ? (fold_ups ? (L + 1) * iterations : L * (1 + iterations) + iterations) : L;
    int cur = flips & 1;

    // ---- level overlap (opt-in experiment, NSOF_OVERLAP=1) -----------------------------------------------------------
    // The iteration and expansion kernels hold a CU through its LDS (159 / 41 KB per workgroup) while the counters show
    // VALU and HBM only 50-65 % busy; the pyramid-level and flow-resample kernels use no LDS and few registers.  On a
    // side stream, the pyramid level of level k-1 runs next to the iterations of level k and the flow resample for
    // level k-1 next to its polynomial expansion; events order the hand-overs.  Same kernels, arguments and results.
    // Measured (256 pairs 1080p): every kernel slows down by what the others gain -- iterate 23.7 -> 26.3 ms, polyexp
    // 8.9 -> 10.4, prep 2.9 -> 3.9, resample 1.5 -> 2.3 per step, 6877 vs 6804 pairs/s -- there is no idle capacity to
    // harvest next to these kernels, so the single-stream order stays the default.
#ifdef NSOF_AB
    static const bool overlap_env = [] { const char* e = NSOF_AB_GETENV("NSOF_OVERLAP"); return e && e[0] == '1'; }();
#endif
#ifdef NSOF_AB   // tuning builds only: NSOF_OVERLAP=1 (level overlap on a side stream; measured no faster)
    if (fused && !exact && !fold_ups && overlap_env && L >= 1 && iterations > 0) {
        if (!ctx->side) NSOF_HIP(ctx, hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking));
        while (ctx->ov_events.size() < (size_t)4 * (L + 1)) {
            hipEvent_t ev;
            NSOF_HIP(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            ctx->ov_events.push_back(ev);
        }
        auto EV = [&](int k, int which) { return ctx->ov_events[(size_t)4 * k + which]; };   // 0 prep, 1 poly, 2 iter, 3 ups
        struct StreamSwap {
            nsof_ctx* c; hipStream_t saved;
            StreamSwap(nsof_ctx* cc, hipStream_t s) : c(cc), saved(cc->stream) { c->stream = s; }
            ~StreamSwap() { c->stream = saved; }
        };
        // second level-image buffer behind the regular workspace
        const size_t base_bytes = szI + szR + szS + szM + szV;
        if ((rc = nsof_ws_reserve(ctx, &ctx->ws, &ctx->ws_bytes, base_bytes + szI))) return rc;
        base = (char*)ctx->ws;
        dI = (float*)base;
        dR = (float*)(base + szI);
        dS = (float*)(base + szI + szR);
        fb[1] = dS;
        float* Ibuf[2] = {dI, (float*)(base + base_bytes)};
        auto prep_level = [&](int k, float* I) -> int {
            int wk, hk, ks;
            double sg;
            nsof_farneback_level_size(width, height, pyr_scale, k, &wk, &hk, &ks, &sg);
            nsof_blur_taps bt;
            if (int r = nsof_host_blur_taps(ks, sg, &bt))
                return nsof_set_error(ctx, r, "pyramid blur kernel size %d unsupported (max %d)", ks, NSOF_MAX_BLUR_TAPS - 1);
            const size_t nk = (size_t)wk * hk;
            if (sequence) return NSOF_PYR_SEL(ctx, nsof_launch_prep, (int)n_img, d_prev, row_stride, pair_stride, width, height, wk, hk, bt, I);
            for (int i = 0; i < 2; i++)
                if (int r = NSOF_PYR_SEL(ctx, nsof_launch_prep, n_pairs, i == 0 ? d_prev : d_next, row_stride, pair_stride, width, height, wk,
                                             hk, bt, I + (size_t)i * B * nk))
                    return r;
            return NSOF_OK;
        };
        const hipStream_t mainS = ctx->stream, sideS = ctx->side;
        if ((rc = prep_level(L, Ibuf[L & 1]))) return rc;
        for (int k = L; k >= 0; k--) {
            int wk, hk;
            nsof_farneback_level_size(width, height, pyr_scale, k, &wk, &hk, nullptr, nullptr);
            const size_t nk = (size_t)wk * hk;
            if (k < L) NSOF_HIP(ctx, hipStreamWaitEvent(mainS, EV(k, 0), 0));          // this level's images are ready
            if ((rc = nsof_launch_polyexp(ctx, (int)n_img, Ibuf[k & 1], wk, hk, ptaps, dR))) return rc;
            if (k > 0) {   // next level's images: on the side stream, next to this level's iterations
                NSOF_HIP(ctx, hipEventRecord(EV(k, 1), mainS));
                NSOF_HIP(ctx, hipStreamWaitEvent(sideS, EV(k, 1), 0));
                StreamSwap sw(ctx, sideS);
                if ((rc = prep_level(k - 1, Ibuf[(k - 1) & 1]))) return rc;
                NSOF_HIP(ctx, hipEventRecord(EV(k - 1, 0), sideS));
            }
            if (k == L) NSOF_HIP(ctx, hipMemsetAsync(fb[cur], 0, B * nk * 8, mainS));
            else NSOF_HIP(ctx, hipStreamWaitEvent(mainS, EV(k, 3), 0));                // the resampled flow is ready
            const float* R0 = dR;
            const float* R1 = dR + (sequence ? (size_t)1 : B) * 5 * nk;
            for (int it = 0; it < iterations; it++) {
                if ((rc = nsof_launch_iterate(ctx, n_pairs, R0, R1, 5 * nk, fb[cur], fb[cur ^ 1], wk, hk, winsize))) return rc;
                cur ^= 1;
            }
            if (k > 0) {   // resample this level's flow for the next one: on the side stream, next to its expansion
                int w1, h1;
                nsof_farneback_level_size(width, height, pyr_scale, k - 1, &w1, &h1, nullptr, nullptr);
                NSOF_HIP(ctx, hipEventRecord(EV(k, 2), mainS));
                NSOF_HIP(ctx, hipStreamWaitEvent(sideS, EV(k, 2), 0));
                StreamSwap sw(ctx, sideS);
                if ((rc = NSOF_PYR_SEL(ctx, nsof_launch_flow_upsample, n_pairs, fb[cur], wk, hk, fb[cur ^ 1], w1, h1, (float)(1. / pyr_scale))))
                    return rc;
                cur ^= 1;
                NSOF_HIP(ctx, hipEventRecord(EV(k - 1, 3), sideS));
            }
        }
        if (fb[cur] != d_flow)
            NSOF_HIP(ctx, hipMemcpyAsync(d_flow, fb[cur], B * n0 * 8, hipMemcpyDeviceToDevice, ctx->stream));
        return NSOF_OK;
    }
#endif

    // prev and next frames of a batch that lie back to back (the host-pointer entry stages a lone pair that way) are one
    // array of 2 B images: one pyramid launch per level instead of two (a lone call's launches have a ~5 us floor each)
    const bool prep_merged = !sequence && d_next == d_prev + (ptrdiff_t)n_pairs * pair_stride;
    auto prep_level = [&](int wk, int hk, const nsof_blur_taps& bt, float* I) -> int {
        const size_t nk = (size_t)wk * hk;
        if (sequence || prep_merged)
            return NSOF_PYR_SEL(ctx, nsof_launch_prep, (int)n_img, d_prev, row_stride, pair_stride, width, height, wk, hk, bt, I);
        for (int i = 0; i < 2; i++)
            if (int r = NSOF_PYR_SEL(ctx, nsof_launch_prep, n_pairs, i == 0 ? d_prev : d_next, row_stride, pair_stride, width, height,
                                     wk, hk, bt, I + (size_t)i * B * nk))
                return r;
        return NSOF_OK;
    };

    // Level 0 (the frame's own size, 3-tap smoothing): the expansion kernel forms the level image itself from the 8-bit
    // frames (k_polyexp_rs<.., U8>): no pyramid launch, no image written and read back.  Not with the FMA twin of the
    // pyramid stages nor with the float expansion (their kernels have no such form).
    static const bool poly_u8_off = [] { const char* e = NSOF_AB_GETENV("NSOF_POLY_U8"); return e && e[0] == '0'; }();
    const bool poly_u8 = !poly_u8_off && !ctx->opt_pyr_fma && !ctx->opt_polyexp_f32 && width >= 2 && height >= 2;
    float* Ifused[4] = {nullptr, nullptr, nullptr, nullptr};   // level images already made by the three-level launch
    auto level_expansion = [&](int k, int wk, int hk, const nsof_blur_taps& bt, float* I, float* Rk) -> int {
        if (k >= 1 && k <= 3 && Ifused[k]) return nsof_launch_polyexp(ctx, (int)n_img, Ifused[k], wk, hk, ptaps, Rk);
        if (poly_u8 && k == 0 && bt.ksize == 3 && wk == width && hk == height) {
            const bool one = sequence || prep_merged;
            return nsof_launch_polyexp_u8(ctx, (int)n_img, d_prev, one ? d_prev : d_next, one ? (int)n_img : n_pairs, row_stride,
                                          pair_stride, width, height, ptaps, bt.k[1], bt.k[2], Rk);
        }
        if (int r = prep_level(wk, hk, bt, I)) return r;
        return nsof_launch_polyexp(ctx, (int)n_img, I, wk, hk, ptaps, Rk);
    };

    // ---- small batches (the three-kernel exact form): the latency schedule ---------------------------------------------
    // A lone call is a chain of ~50 launches that each use a fraction of the chip and cost >= ~5 us (profiles/
    // r03_lone_call_timeline.txt: 762 us at 1080p, a third of it in the two coarsest levels).  Only the FLOW couples the
    // levels; pyramid level and polynomial expansion of every level depend on the input frames alone.  So they move to a
    // side stream (levels L-1 .. 0, into per-level buffers) and run next to the iterations of the coarser levels on the
    // main stream; an event per level hands the expansion over.  Same kernels, same arguments, same bits.
    if (exact_lat && fused && L >= 1 && iterations > 0) {
        std::vector<size_t> offI(L + 1), offR(L + 1);
        size_t totI = 0, totR = 0;
        for (int k = 0; k <= L; k++) {
            int wk, hk;
            nsof_farneback_level_size(width, height, pyr_scale, k, &wk, &hk, nullptr, nullptr);
            offI[k] = totI; offR[k] = totR;
            totI += align_up(n_img * (size_t)wk * hk * 4, 256);
            totR += align_up(n_img * 5 * (size_t)wk * hk * 4, 256);
        }
        if ((rc = nsof_ws_reserve(ctx, &ctx->ws, &ctx->ws_bytes, totI + totR + szS + szM + szV))) return rc;
        base = (char*)ctx->ws;
        float* dS2 = (float*)(base + totI + totR);
        float* dM2 = (float*)(base + totI + totR + szS);
        double* dV2 = (double*)(base + totI + totR + szS + szM);
        float* fl[2] = {d_flow, dS2};
        int c = (L * (1 + iterations) + iterations) & 1;
        if (!ctx->side) NSOF_HIP(ctx, hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking));
        while (ctx->ov_events.size() < (size_t)(L + 2)) {
            hipEvent_t ev;
            NSOF_HIP(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            ctx->ov_events.push_back(ev);
        }
        struct StreamSwap {
            nsof_ctx* cx; hipStream_t saved;
            StreamSwap(nsof_ctx* cc, hipStream_t st) : cx(cc), saved(cc->stream) { cx->stream = st; }
            ~StreamSwap() { cx->stream = saved; }
        };
        auto level_images = [&](int k) -> int {   // pyramid level + expansion of level k on the CURRENT ctx->stream
            int wk, hk, ks;
            double sg;
            nsof_farneback_level_size(width, height, pyr_scale, k, &wk, &hk, &ks, &sg);
            nsof_blur_taps bt;
            if (int r = nsof_host_blur_taps(ks, sg, &bt))
                return nsof_set_error(ctx, r, "pyramid blur kernel size %d unsupported (max %d)", ks, NSOF_MAX_BLUR_TAPS - 1);
            return level_expansion(k, wk, hk, bt, (float*)(base + offI[k]), (float*)(base + totI + offR[k]));
        };
        const hipStream_t mainS = ctx->stream, sideS = ctx->side;
        NSOF_HIP(ctx, hipEventRecord(ctx->ov_events[L + 1], mainS));            // the frames are on the device; earlier calls are done
        NSOF_HIP(ctx, hipStreamWaitEvent(sideS, ctx->ov_events[L + 1], 0));
        if ((rc = level_images(L))) return rc;                                  // the coarsest level: needed first, main stream
        {
            StreamSwap sw(ctx, sideS);
            for (int k = L - 1; k >= 0; k--) {
                if ((rc = level_images(k))) return rc;
                NSOF_HIP(ctx, hipEventRecord(ctx->ov_events[k], sideS));
            }
        }
        int pw2 = 0, ph2 = 0;
        for (int k = L; k >= 0; k--) {
            int wk, hk;
            nsof_farneback_level_size(width, height, pyr_scale, k, &wk, &hk, nullptr, nullptr);
            const size_t nk = (size_t)wk * hk;
            if (k == L) {
                NSOF_HIP(ctx, hipMemsetAsync(fl[c], 0, B * nk * 8, mainS));
            } else {
                if ((rc = NSOF_PYR_SEL(ctx, nsof_launch_flow_upsample, n_pairs, fl[c], pw2, ph2, fl[c ^ 1], wk, hk, (float)(1. / pyr_scale))))
                    return rc;
                c ^= 1;
                NSOF_HIP(ctx, hipStreamWaitEvent(mainS, ctx->ov_events[k], 0));   // this level's expansion is ready
            }
            const float* R0 = (const float*)(base + totI + offR[k]);
            const float* R1 = R0 + (sequence ? (size_t)1 : B) * 5 * nk;
            for (int it = 0; it < iterations; it++) {
                if ((rc = nsof_launch_iterate_lat(ctx, n_pairs, R0, R1, 5 * nk, fl[c], fl[c ^ 1], wk, hk, winsize, dM2, dV2))) return rc;
                c ^= 1;
            }
            pw2 = wk;
            ph2 = hk;
        }
        if (fl[c] != d_flow)
            NSOF_HIP(ctx, hipMemcpyAsync(d_flow, fl[c], B * n0 * 8, hipMemcpyDeviceToDevice, mainS));
        return NSOF_OK;
    }

    // pyr_scale 0.5 with three coarser levels (the headline configuration): levels 1..3 smooth and decimate the same
    // full-resolution frames -- one launch makes all three (k_prep_decim3), into the level-image buffer that level 0 no
    // longer needs before the coarser levels are done with it
    static const bool decim3_off = [] { const char* e = NSOF_AB_GETENV("NSOF_DECIM3"); return e && e[0] == '0'; }();
    if (L == 3 && !decim3_off) {
        nsof_blur_taps bt3[3];
        size_t nk3[3];
        bool exact3 = true;
        for (int k = 1; k <= 3 && exact3; k++) {
            int wk, hk, ks;
            double sg;
            nsof_farneback_level_size(width, height, pyr_scale, k, &wk, &hk, &ks, &sg);
            exact3 = wk * (1 << k) == width && hk * (1 << k) == height && nsof_host_blur_taps(ks, sg, &bt3[k - 1]) == 0;
            nk3[k - 1] = (size_t)wk * hk;
        }
        if (exact3) {
            float* I3[3] = {dI, dI + n_img * nk3[0], dI + n_img * (nk3[0] + nk3[1])};
            const bool one = sequence || prep_merged;
            rc = NSOF_PYR_SEL(ctx, nsof_launch_prep_decim3, one ? (int)n_img : n_pairs, d_prev, row_stride, pair_stride, width, height, bt3, I3);
            if (rc == NSOF_OK && !one) {
                float* I3n[3] = {I3[0] + B * nk3[0], I3[1] + B * nk3[1], I3[2] + B * nk3[2]};
                rc = NSOF_PYR_SEL(ctx, nsof_launch_prep_decim3, n_pairs, d_next, row_stride, pair_stride, width, height, bt3, I3n);
            }
            if (rc == NSOF_OK) {
                for (int k = 1; k <= 3; k++) Ifused[k] = I3[k - 1];
            } else if (rc != NSOF_EUNSUPPORTED) {
                return rc;
            }
        }
    }

    bool have_prev = false;
    int pw = 0, ph = 0;
    for (int k = L; k >= 0; k--) {
        int wk, hk, ks;
        double sg;
        nsof_farneback_level_size(width, height, pyr_scale, k, &wk, &hk, &ks, &sg);
        nsof_blur_taps btaps;
        if ((rc = nsof_host_blur_taps(ks, sg, &btaps)))
            return nsof_set_error(ctx, rc, "pyramid blur kernel size %d unsupported (max %d)", ks,
                                  NSOF_MAX_BLUR_TAPS - 1);
        const size_t nk = (size_t)wk * hk;
        bool pending_ups = false;
        if (!have_prev) {
            NSOF_HIP(ctx, hipMemsetAsync(fb[cur], 0, B * nk * 8, ctx->stream));
        } else if (fold_ups) {
            pending_ups = true;   // fb[cur] still holds the coarse flow (pw x ph)
        } else {
            if ((rc = NSOF_PYR_SEL(ctx, nsof_launch_flow_upsample, n_pairs, fb[cur], pw, ph, fb[cur ^ 1], wk, hk,
                                                (float)(1. / pyr_scale))))
                return rc;
            cur ^= 1;
        }
        // image-major: dI [n_img][hk][wk], dR [n_img][5*hk*wk].  Pairs: all prev frames then all next frames
        // (R1 = R0 + B images); sequence: the frames in order (R1 = R0 + 1 image).
        if ((rc = level_expansion(k, wk, hk, btaps, dI, dR))) return rc;
        const float* R0 = dR;
        const float* R1 = dR + (sequence ? (size_t)1 : B) * 5 * nk;
        if (fused) {
            for (int it = 0; it < iterations; it++) {
                if (it == 0 && pending_ups)
                    rc = nsof_launch_iterate_upsample(ctx, n_pairs, R0, R1, 5 * nk, fb[cur], pw, ph,
                                                      (float)(1. / pyr_scale), fb[cur ^ 1], wk, hk, winsize);
                else if (exact_lat)
                    rc = nsof_launch_iterate_lat(ctx, n_pairs, R0, R1, 5 * nk, fb[cur], fb[cur ^ 1], wk, hk, winsize, dM, dV);
                else if (exact_x)
                    rc = nsof_launch_iterate_x(ctx, n_pairs, R0, R1, 5 * nk, fb[cur], fb[cur ^ 1], wk, hk, winsize);
                else if (exact_fused)
                    rc = nsof_launch_iterate_exact(ctx, n_pairs, R0, R1, 5 * nk, fb[cur], fb[cur ^ 1], wk, hk, winsize, dV);
                else
                    rc = nsof_launch_iterate(ctx, n_pairs, R0, R1, 5 * nk, fb[cur], fb[cur ^ 1], wk, hk, winsize);
                if (rc) return rc;
                cur ^= 1;
            }
        } else {
            float* flow = fb[cur];
            if ((rc = nsof_launch_update_matrices(ctx, n_pairs, R0, R1, 5 * nk, flow, wk, hk, dM))) return rc;
            for (int it = 0; it < iterations; it++) {
                if (exact) rc = nsof_launch_blur_solve_exact(ctx, n_pairs, dM, wk, hk, winsize, dV, flow);
                else rc = nsof_launch_blur_solve(ctx, n_pairs, dM, wk, hk, winsize, flow);
                if (rc) return rc;
                if (it < iterations - 1)
                    if ((rc = nsof_launch_update_matrices(ctx, n_pairs, R0, R1, 5 * nk, flow, wk, hk, dM))) return rc;
            }
        }
        have_prev = true;
        pw = wk;
        ph = hk;
    }
    if (fb[cur] != d_flow)  // cannot happen by construction; keep the result correct regardless
        NSOF_HIP(ctx, hipMemcpyAsync(d_flow, fb[cur], B * n0 * 8, hipMemcpyDeviceToDevice, ctx->stream));
    return NSOF_OK;
}

extern "C" int nsof_farneback_u8_batch_dev(nsof_ctx* ctx, int n_pairs, const uint8_t* d_prev, const uint8_t* d_next,
                                           ptrdiff_t row_stride, ptrdiff_t pair_stride, int width, int height,
                                           float* d_flow, double pyr_scale, int levels, int winsize, int iterations,
                                           int poly_n, double poly_sigma, int flags)
{
    return nsof_farneback_core(ctx, false, n_pairs, d_prev, d_next, row_stride, pair_stride, width, height, d_flow, pyr_scale,
                          levels, winsize, iterations, poly_n, poly_sigma, flags);
}

extern "C" int nsof_farneback_u8_sequence_dev(nsof_ctx* ctx, int n_frames, const uint8_t* d_frames,
                                              ptrdiff_t row_stride, ptrdiff_t frame_stride, int width, int height,
                                              float* d_flow, double pyr_scale, int levels, int winsize,
                                              int iterations, int poly_n, double poly_sigma, int flags)
{
    if (!ctx) return NSOF_EINVAL;
    if (n_frames < 2) return nsof_set_error(ctx, NSOF_EINVAL, "a sequence needs at least 2 frames");
    return nsof_farneback_core(ctx, true, n_frames - 1, d_frames, nullptr, row_stride, frame_stride, width, height, d_flow,
                          pyr_scale, levels, winsize, iterations, poly_n, poly_sigma, flags);
}

extern "C" int nsof_farneback_u8(nsof_ctx* ctx, const uint8_t* prev, ptrdiff_t prev_stride, const uint8_t* next,
                                 ptrdiff_t next_stride, int width, int height, float* flow, ptrdiff_t flow_stride,
                                 double pyr_scale, int levels, int winsize, int iterations, int poly_n,
                                 double poly_sigma, int flags)
{
    if (!ctx) return NSOF_EINVAL;
    if (!prev || !next || !flow) return nsof_set_error(ctx, NSOF_EINVAL, "null image pointer");
    int rc = nsof_check_farneback_params(ctx, width, height, pyr_scale, levels, winsize, iterations, poly_n, flags);
    if (rc) return rc;
    if (flow_stride < (ptrdiff_t)(width * 8)) return nsof_set_error(ctx, NSOF_EINVAL, "flow_stride < width*8");
    NSOF_HIP(ctx, hipSetDevice(ctx->device));
    // Strided host views are packed row by row into a pinned staging buffer and moved with ONE linear copy per
    // direction: hipMemcpy2D degenerates to a copy per row for widths that are not nicely aligned (measured 12 ms
    // for an 801x801 pair against 3 ms of kernels).
    const size_t n0 = (size_t)width * height, pitch = (size_t)width;
    const size_t szU = align_up(n0, 256), szF = align_up(n0 * 8, 256);
    if ((rc = nsof_ws_reserve(ctx, &ctx->stage, &ctx->stage_bytes, 2 * szU + szF))) return rc;
    if ((rc = nsof_hstage_reserve(ctx, 2 * szU + szF))) return rc;
    uint8_t* hP = (uint8_t*)ctx->hstage;
    uint8_t* hN = hP + szU;
    float* hF = (float*)(hN + szU);
    uint8_t* dP = (uint8_t*)ctx->stage;
    uint8_t* dN = dP + szU;
    float* dFl = (float*)(dN + szU);
    const bool in_dense = prev_stride == (ptrdiff_t)width && next_stride == (ptrdiff_t)width;
    const bool out_dense = flow_stride == (ptrdiff_t)width * 8;
    if (in_dense) {   // contiguous frames: straight from the caller's memory
        NSOF_HIP(ctx, hipMemcpyAsync(dP, prev, n0, hipMemcpyHostToDevice, ctx->stream));
        NSOF_HIP(ctx, hipMemcpyAsync(dN, next, n0, hipMemcpyHostToDevice, ctx->stream));
    } else {
        for (int y = 0; y < height; y++) {
            memcpy(hP + (size_t)y * width, prev + (ptrdiff_t)y * prev_stride, (size_t)width);
            memcpy(hN + (size_t)y * width, next + (ptrdiff_t)y * next_stride, (size_t)width);
        }
        NSOF_HIP(ctx, hipMemcpyAsync(dP, hP, 2 * szU, hipMemcpyHostToDevice, ctx->stream));
    }
    rc = nsof_farneback_u8_batch_dev(ctx, 1, dP, dN, (ptrdiff_t)pitch, (ptrdiff_t)szU, width, height, dFl, pyr_scale,
                                     levels, winsize, iterations, poly_n, poly_sigma, flags);
    if (rc) return rc;
    NSOF_HIP(ctx, hipMemcpyAsync(out_dense ? flow : hF, dFl, n0 * 8, hipMemcpyDeviceToHost, ctx->stream));
    // a hand-over between workgroups that never arrived (the exact-order kernels' bounded waits) fails the call, as cv2
    // raises where it fails: the flow of such a launch is never handed back as a result
    if ((rc = nsof_stream_sync_checked(ctx))) return rc;
    if (!out_dense)
        for (int y = 0; y < height; y++)
            memcpy((char*)flow + (ptrdiff_t)y * flow_stride, hF + (size_t)y * width * 2, (size_t)width * 8);
    return NSOF_OK;
}

// ---- ROI gating (host arithmetic on maps of at most a few hundred cells) -------------------------------------------
extern "C" int nsof_roi_from_surface(const double* current, int rows, int cols, int frame_w, int frame_h, int memsize,
                                     int thres, int extend_left, int extend_right, int extend_upper, int extend_lower,
                                     int connectivity, int flag, int* rects, int max_rects)
{
    if (!current || rows < 1 || cols < 1 || frame_w < 1 || frame_h < 1 || memsize < 1 || (connectivity != 4 && connectivity != 8) ||
        (flag != 1 && flag != 2) || max_rects < 0 || (max_rects > 0 && !rects))
        return NSOF_EINVAL;
    const int th = frame_h / memsize, tw = frame_w / memsize;   // the transition picture: int(h / MS) x int(w / MS)
    if (rows > th || cols > tw) return NSOF_ESHAPE;             // the reference's numba loop would write out of bounds
    std::vector<int> lab((size_t)th * tw, 0);
    std::vector<unsigned char> on((size_t)th * tw, 0);
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) {
            double g = -3366.0 / std::log10(current[(size_t)y * cols + x]) - 306.0;
            g = g < 0.0 ? 0.0 : (g > 255.0 ? 255.0 : g);        // NaN (I <= 0) compares false twice and casts to 0
            const int gi = (g == g) ? (int)(unsigned char)g : 0;
            on[(size_t)y * tw + x] = gi >= thres;
        }
    struct Box { int x0, y0, x1, y1; };
    std::vector<Box> boxes;
    std::vector<int> stack;
    for (int y = 0; y < th; y++)
        for (int x = 0; x < tw; x++) {
            if (!on[(size_t)y * tw + x] || lab[(size_t)y * tw + x]) continue;
            boxes.push_back({x, y, x, y});
            const int id = (int)boxes.size();
            lab[(size_t)y * tw + x] = id;
            stack.assign(1, y * tw + x);
            while (!stack.empty()) {
                const int p = stack.back();
                stack.pop_back();
                const int py = p / tw, px = p % tw;
                Box& b = boxes[id - 1];
                b.x0 = px < b.x0 ? px : b.x0; b.x1 = px > b.x1 ? px : b.x1;
                b.y0 = py < b.y0 ? py : b.y0; b.y1 = py > b.y1 ? py : b.y1;
                for (int dy = -1; dy <= 1; dy++)
                    for (int dx = -1; dx <= 1; dx++) {
                        if ((!dx && !dy) || (connectivity == 4 && dx && dy)) continue;
                        const int ny = py + dy, nx = px + dx;
                        if (ny < 0 || ny >= th || nx < 0 || nx >= tw) continue;
                        const size_t q = (size_t)ny * tw + nx;
                        if (on[q] && !lab[q]) { lab[q] = id; stack.push_back((int)q); }
                    }
            }
        }
    if (boxes.empty()) return 0;
    if (flag == 2) {   // union box of all components
        Box u = boxes[0];
        for (const Box& b : boxes) {
            u.x0 = b.x0 < u.x0 ? b.x0 : u.x0; u.y0 = b.y0 < u.y0 ? b.y0 : u.y0;
            u.x1 = b.x1 > u.x1 ? b.x1 : u.x1; u.y1 = b.y1 > u.y1 ? b.y1 : u.y1;
        }
        boxes.assign(1, u);
    }
    int n = 0;
    for (const Box& b : boxes) {
        const int x0 = std::max(b.x0 * memsize - extend_left, 0), y0 = std::max(b.y0 * memsize - extend_upper, 0);
        const int x1 = std::min((b.x1 + 1) * memsize + extend_right, frame_w), y1 = std::min((b.y1 + 1) * memsize + extend_lower, frame_h);
        if (n < max_rects) { rects[4 * n] = x0; rects[4 * n + 1] = y0; rects[4 * n + 2] = x1; rects[4 * n + 3] = y1; }
        n++;
    }
    return n;
}
