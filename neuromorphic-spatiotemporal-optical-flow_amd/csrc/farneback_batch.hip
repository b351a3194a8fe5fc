// Shape-heterogeneous Farneback batches: the work-list driver and the pipelined host entry point.
//
// The reference's gated path calls cv2.calcOpticalFlowFarneback once per ROI, on crops whose shape changes from
// call to call (/root/reference/optical_flow_seg.py:129-164 per connected component, :186-203 union box), and its
// evaluation loop mixes those with full-frame calls (:492-496).  A crop of 520x200 px cannot fill 256 CUs -- the
// fused iteration walks its rows serially in 3 workgroups -- so here MANY such calls share every launch:
//
//   * nsof_farneback_u8_batch_desc_dev: per level one launch per stage over a device table of work items
//     (nsof_het_item, nsof_internal.h); an item takes part from its own coarsest level on; the last iteration of
//     level 0 writes straight into the caller's (strided) flow field, so an ROI result lands in the frame-sized
//     canvas without a paste.  Arithmetic per item is that of the per-call path, bit for bit.
//   * nsof_farneback_u8_batch: the same for HOST memory, as a three-stage pipeline over chunks of the list
//     (upload of chunk c+1 and download of chunk c-1 on their own streams while chunk c computes).
#include <sched.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#include "nsof_internal.h"

namespace {

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct Params {
    double pyr_scale;
    int levels, winsize, iterations, poly_n;
    double poly_sigma;
    int flags;
};

int validate_desc(nsof_ctx* ctx, int i, const nsof_pair_desc& d, const Params& p)
{
    if (!d.prev || !d.next || !d.flow) return nsof_set_error(ctx, NSOF_EINVAL, "pair %d: null pointer", i);
    int rc = nsof_check_farneback_params(ctx, d.width, d.height, p.pyr_scale, p.levels, p.winsize, p.iterations, p.poly_n,
                                         p.flags);
    if (rc) return rc;
    if (d.prev_stride < d.width || d.next_stride < d.width)
        return nsof_set_error(ctx, NSOF_EINVAL, "pair %d: row stride < width", i);
    if (d.flow_stride < (ptrdiff_t)d.width * 8 || (d.flow_stride & 7) || (reinterpret_cast<uintptr_t>(d.flow) & 7))
        return nsof_set_error(ctx, NSOF_EINVAL, "pair %d: flow stride %lld / pointer must be multiples of 8 bytes and "
                              "the stride >= width*8", i, (long long)d.flow_stride);
    return NSOF_OK;
}

// One item through the uniform-shape driver (shapes or parameters the work-list kernels do not cover): frames are
// packed densely, the result is copied into the caller's strided field.
int fallback_item(nsof_ctx* ctx, const nsof_pair_desc& d, const Params& p)
{
    const size_t n0 = (size_t)d.width * d.height;
    const size_t szU = align_up(n0, 256), szF = align_up(n0 * 8, 256);
    int rc = nsof_ws_reserve(ctx, &ctx->stage, &ctx->stage_bytes, 2 * szU + szF);
    if (rc) return rc;
    uint8_t* dP = (uint8_t*)ctx->stage;
    uint8_t* dN = dP + szU;
    float* dF = (float*)(dN + szU);
    NSOF_HIP(ctx, hipMemcpy2DAsync(dP, d.width, d.prev, d.prev_stride, d.width, d.height, hipMemcpyDeviceToDevice, ctx->stream));
    NSOF_HIP(ctx, hipMemcpy2DAsync(dN, d.width, d.next, d.next_stride, d.width, d.height, hipMemcpyDeviceToDevice, ctx->stream));
    rc = nsof_farneback_core(ctx, false, 1, dP, dN, d.width, (ptrdiff_t)szU, d.width, d.height, dF, p.pyr_scale, p.levels,
                             p.winsize, p.iterations, p.poly_n, p.poly_sigma, p.flags);
    if (rc) return rc;
    NSOF_HIP(ctx, hipMemcpy2DAsync(d.flow, d.flow_stride, dF, (size_t)d.width * 8, (size_t)d.width * 8, d.height,
                                   hipMemcpyDeviceToDevice, ctx->stream));
    return NSOF_OK;
}

// A size class of a level's item table: items [start, start + count) with extents up to max_w x max_h.
struct HetClass {
    int start, count, max_w, max_h;
};

// Sorts a level's items into size classes (power-of-two buckets of width and height).  The tiled kernels of a level
// launch a grid over the LARGEST extents of the items they are given and workgroups outside an item leave at once; with
// one full frame among thousands of small crops nearly every workgroup of such a launch would be one of those (measured:
// 1.9 of 2.0 ms of a level-0 launch), so each class gets its own launch.  Classes with few items share one catch-all.
void sort_into_classes(nsof_het_item* t, int n, std::vector<std::pair<int, int>>& keyed, std::vector<nsof_het_item>& sorted,
                       std::vector<HetClass>& out)
{
    out.clear();
    if (n == 0) return;
    auto bucket = [](int v, int first) {
        int b = 0;
        while (b < 7 && v > (first << b)) b++;
        return b;
    };
    int count[64] = {0};
    keyed.resize(n);
    for (int i = 0; i < n; i++) {
        const int key = bucket(t[i].hk, 16) * 8 + bucket(t[i].wk, 64);
        keyed[i] = {key, i};
        count[key]++;
    }
    // own launch: the 6 most populated buckets with at least 16 items; the rest -> catch-all (key 64)
    int order[64];
    for (int i = 0; i < 64; i++) order[i] = i;
    std::stable_sort(order, order + 64, [&](int a, int b) { return count[a] > count[b]; });
    bool own[64] = {false};
    for (int r = 0; r < 6; r++) own[order[r]] = count[order[r]] >= 16;
    for (auto& kv : keyed)
        if (!own[kv.first]) kv.first = 64;
    std::stable_sort(keyed.begin(), keyed.end(), [](const std::pair<int, int>& a, const std::pair<int, int>& b) { return a.first < b.first; });
    sorted.resize(n);
    for (int i = 0; i < n; i++) sorted[i] = t[keyed[i].second];
    memcpy(t, sorted.data(), (size_t)n * sizeof(nsof_het_item));
    for (int i = 0; i < n; i++) {
        if (i == 0 || keyed[i].first != keyed[i - 1].first) out.push_back({i, 0, 0, 0});
        HetClass& c = out.back();
        c.count++;
        c.max_w = std::max(c.max_w, t[i].wk);
        c.max_h = std::max(c.max_h, t[i].hk);
    }
}

// The fused iteration kernel's job table of one level (layout: k_iterate_x; *stride = the longest list).  Items go to the
// XCD list with the least work so far, tallest first, the strips of an item one after the other (a strip waits for its
// left neighbour's carries, which must therefore have been taken earlier from the same list).  Returns the number of
// jobs; the table takes 8 + 8 * *stride words.
int build_xjobs(const nsof_het_item* t, int n, unsigned* xj, int* stride, std::vector<std::pair<int, int>>& order,
                std::vector<unsigned> (&lists)[8])
{
    order.resize(n);
    for (int i = 0; i < n; i++) order[i] = {t[i].hk, i};
    std::stable_sort(order.begin(), order.end(), [](const std::pair<int, int>& a, const std::pair<int, int>& b) { return a.first > b.first; });
    long long load[8] = {0};
    for (auto& l : lists) l.clear();
    int jobs = 0;
    for (int r = 0; r < n; r++) {
        const int i = order[r].second;
        const int strips = (t[i].wk + NSOF_X_STRIP - 1) / NSOF_X_STRIP;
        int x = 0;
        for (int k = 1; k < 8; k++)
            if (load[k] < load[x]) x = k;
        for (int s = 0; s < strips; s++) lists[x].push_back((unsigned)i << 8 | (unsigned)s);
        load[x] += (long long)strips * ((t[i].hk + 3) / 4 + 12);   // steps of the walk + a job's start-up
        jobs += strips;
    }
    size_t longest = 1;
    for (auto& l : lists) longest = std::max(longest, l.size());
    for (int k = 0; k < 8; k++) {
        xj[k] = (unsigned)lists[k].size();
        if (!lists[k].empty()) memcpy(xj + 8 + (size_t)k * longest, lists[k].data(), lists[k].size() * sizeof(unsigned));
    }
    *stride = (int)longest;
    return jobs;
}

// The work-list driver.  descs: HOST array whose pointers are DEVICE addresses.
int het_core(nsof_ctx* ctx, int n, const nsof_pair_desc* descs, const Params& p)
{
    NSOF_HIP(ctx, hipSetDevice(ctx->device));
    for (int i = 0; i < n; i++)
        if (int rc = validate_desc(ctx, i, descs[i], p)) return rc;

    // A list of equal shapes laid out at constant strides (what the pipelined host entry builds for a video) is the
    // uniform batch: it takes that driver and its specialised kernels (decimating pyramid levels, XCD-aware placement).
    if (n >= 2) {
        const nsof_pair_desc& d0 = descs[0];
        const ptrdiff_t ps = descs[1].prev - d0.prev;
        bool uniform = d0.prev_stride == d0.next_stride && ps > 0 && d0.flow_stride == (ptrdiff_t)d0.width * 8;
        const ptrdiff_t fs = (ptrdiff_t)d0.width * d0.height * 2;   // floats between consecutive dense flow fields
        for (int i = 1; i < n && uniform; i++) {
            const nsof_pair_desc& d = descs[i];
            uniform = d.width == d0.width && d.height == d0.height && d.prev_stride == d0.prev_stride &&
                      d.next_stride == d0.prev_stride && d.flow_stride == d0.flow_stride &&
                      d.prev - d0.prev == (ptrdiff_t)i * ps && d.next - d0.next == (ptrdiff_t)i * ps &&
                      d.flow - d0.flow == (ptrdiff_t)i * fs;
        }
        if (uniform)
            return nsof_farneback_core(ctx, false, n, d0.prev, d0.next, d0.prev_stride, ps, d0.width, d0.height, d0.flow,
                                       p.pyr_scale, p.levels, p.winsize, p.iterations, p.poly_n, p.poly_sigma, p.flags);
    }

    // Items the work-list kernels cover: fused iteration available (window 2..15, >= 1 iteration, at least 2x2 px).
    const bool het_params = p.iterations >= 1 && p.winsize / 2 >= 1 && p.winsize / 2 <= 7;
    const bool exact = ctx->opt_exact_rowsums != 0;
    std::vector<int> het, rest;
    for (int i = 0; i < n; i++)
        (het_params && nsof_iterate_supported(p.winsize, descs[i].width, descs[i].height) ? het : rest).push_back(i);

    if (!het.empty()) {
        const int nh = (int)het.size();
        nsof_poly_taps ptaps;
        int rc = nsof_host_poly_taps(p.poly_n, p.poly_sigma, &ptaps);
        if (rc) return nsof_set_error(ctx, rc, "poly taps");
        // per-item level count; tables per level (items without that level are left out)
        std::vector<int> Li(nh);
        int Lmax = 0;
        for (int j = 0; j < nh; j++) {
            Li[j] = nsof_farneback_effective_levels(descs[het[j]].width, descs[het[j]].height, p.pyr_scale, p.levels);
            Lmax = std::max(Lmax, Li[j]);
        }
        // Per level: the item table (sorted into size classes), then the fused kernel's job table (8 counts + 8 lists).
        long long strips0 = 0;   // strips of the full-resolution level = the most any level has
        for (int j = 0; j < nh; j++) {
            const int st = (descs[het[j]].width + NSOF_X_STRIP - 1) / NSOF_X_STRIP;
            if (st > 255) return nsof_set_error(ctx, NSOF_EUNSUPPORTED, "item %d: width %d above %d", het[j], descs[het[j]].width, 255 * NSOF_X_STRIP);
            strips0 += st;
        }
        if (strips0 >= (1ll << 26) || nh >= (1 << 24))
            return nsof_set_error(ctx, NSOF_EINVAL, "work list too long (%d items, %lld strips)", nh, strips0);
        const size_t xj_words = align_up(8 + 8 * (size_t)strips0, 64);   // words per level at most (every strip in one list)
        const size_t items_bytes = align_up((size_t)(Lmax + 1) * nh * sizeof(nsof_het_item), 256);
        // two table slots used alternately: the upload of call c may still be queued when call c+1 builds its tables
        const size_t tab_bytes = align_up(items_bytes + (size_t)(Lmax + 1) * xj_words * sizeof(unsigned), 256);
        if (ctx->het_bytes < 2 * tab_bytes) {
            NSOF_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (ctx->het_h) hipHostFree(ctx->het_h);
            if (ctx->het_d) hipFree(ctx->het_d);
            ctx->het_h = ctx->het_d = nullptr;
            ctx->het_bytes = 0;
            const size_t cap = align_up(tab_bytes * 4, 4096);
            if (hipHostMalloc(&ctx->het_h, cap, hipHostMallocDefault) != hipSuccess || hipMalloc(&ctx->het_d, cap) != hipSuccess)
                return nsof_set_error(ctx, NSOF_ENOMEM, "work-list tables (%zu bytes)", cap);
            ctx->het_bytes = cap;
        }
        const int slot = ctx->het_flip;
        ctx->het_flip ^= 1;
        if (!ctx->het_ev[slot]) NSOF_HIP(ctx, hipEventCreateWithFlags(&ctx->het_ev[slot], hipEventDisableTiming));
        else NSOF_HIP(ctx, hipEventSynchronize(ctx->het_ev[slot]));   // this slot's previous upload has left the pinned copy
        const size_t slot_off = (size_t)slot * (ctx->het_bytes / 2);
        nsof_het_item* tabs = (nsof_het_item*)((char*)ctx->het_h + slot_off);
        const nsof_het_item* d_tabs = (const nsof_het_item*)((char*)ctx->het_d + slot_off);
        unsigned* xj = (unsigned*)((char*)ctx->het_h + slot_off + items_bytes);
        const unsigned* d_xj = (const unsigned*)((char*)ctx->het_d + slot_off + items_bytes);

        // Build the tables, coarsest level first in memory order k = 0..Lmax (table k at tabs + k*nh).
        std::vector<int> cnt(Lmax + 1, 0), xj_jobs(Lmax + 1, 0), xj_stride(Lmax + 1, 1);
        std::vector<size_t> xj_at(Lmax + 1, 0);   // word offset of level k's job table
        std::vector<unsigned> lists[8];
        size_t xj_used = 0;
        std::vector<unsigned long long> offF_prev(nh, 0);   // the item's flow offset at the next coarser level
        size_t maxI = 0, maxR = 0, maxF = 0;
        std::vector<int> max_w(Lmax + 1, 0), max_h(Lmax + 1, 0);
        std::vector<std::vector<HetClass>> classes(Lmax + 1);
        std::vector<std::pair<int, int>> keyed;   // (class key, position) scratch
        std::vector<nsof_het_item> sorted;
        for (int k = Lmax; k >= 0; k--) {
            unsigned long long oI = 0, oR = 0, oF = 0;
            nsof_het_item* t = tabs + (size_t)k * nh;
            for (int j = 0; j < nh; j++) {
                if (Li[j] < k) continue;
                const nsof_pair_desc& d = descs[het[j]];
                nsof_het_item it;
                memset(&it, 0, sizeof(it));
                it.src[0] = d.prev; it.src[1] = d.next;
                it.src_stride[0] = d.prev_stride; it.src_stride[1] = d.next_stride;
                it.out = d.flow;
                it.out_pitch = d.flow_stride / 8;
                it.W = d.width; it.H = d.height;
                nsof_farneback_level_size(d.width, d.height, p.pyr_scale, k, &it.wk, &it.hk, nullptr, nullptr);
                if (k < Li[j]) nsof_farneback_level_size(d.width, d.height, p.pyr_scale, k + 1, &it.pw, &it.ph, nullptr, nullptr);
                const unsigned long long nk = (unsigned long long)it.wk * it.hk;
                it.offI = oI; it.offR = oR; it.offF = oF; it.offFc = offF_prev[j];
                oI += align_up(2 * nk, 64); oR += align_up(10 * nk, 64); oF += align_up(nk, 32);
                offF_prev[j] = it.offF;
                const bool vec = (d.width & 3) == 0 && d.width >= 8 && (d.prev_stride & 3) == 0 && (d.next_stride & 3) == 0 &&
                                 (reinterpret_cast<uintptr_t>(d.prev) & 3) == 0 && (reinterpret_cast<uintptr_t>(d.next) & 3) == 0;
                it.flags = vec ? NSOF_HET_VEC0 : 0;
                max_w[k] = std::max(max_w[k], it.wk);
                max_h[k] = std::max(max_h[k], it.hk);
                t[cnt[k]++] = it;
            }
            maxI = std::max(maxI, (size_t)oI); maxR = std::max(maxR, (size_t)oR); maxF = std::max(maxF, (size_t)oF);
            sort_into_classes(t, cnt[k], keyed, sorted, classes[k]);
            xj_at[k] = xj_used;
            xj_jobs[k] = build_xjobs(t, cnt[k], xj + xj_used, &xj_stride[k], keyed, lists);
            xj_used += align_up(8 + 8 * (size_t)xj_stride[k], 64);
        }
        NSOF_HIP(ctx, hipMemcpyAsync((void*)d_tabs, tabs, items_bytes + xj_used * sizeof(unsigned), hipMemcpyHostToDevice, ctx->stream));
        NSOF_HIP(ctx, hipEventRecord(ctx->het_ev[slot], ctx->stream));

        // workspace: level images, expansions, two flow buffers (every level uses their leading part)
        const size_t szI = align_up(maxI * 4, 256), szR = align_up(maxR * 4, 256), szF = align_up(maxF * 8, 256);
        static const bool exact_2k = [] { const char* e = NSOF_AB_GETENV("NSOF_EXACT_IMPL"); return e && e[0] == '2'; }();
        const bool exact_x = exact && !exact_2k;   // one fused kernel (k_iterate_x); 2k: column sums through HBM
        // a list too small to fill the chip with (strip, item) jobs: the three-kernel small-batch form of the same order
        long long jobs = 0;
        for (int j = 0; j < nh; j++) {
            jobs += (descs[het[j]].width + 191) / 192;
            if ((unsigned long long)descs[het[j]].width * descs[het[j]].height * 40ull >= (1ull << 32)) jobs = 1ll << 40;   // 32-bit offsets per item
        }
        const bool exact_lat = exact_x && jobs <= ctx->opt_small_batch_jobs;
        const size_t szV = (exact && !exact_x) || exact_lat ? szR : 0;   // column sums, 5 doubles per pixel = the expansion's footprint
        const size_t szM = exact_lat ? align_up(szR / 2, 256) : 0;        // matrices of the small-batch form, 5 floats per pixel
        if ((rc = nsof_ws_reserve(ctx, &ctx->ws, &ctx->ws_bytes, szI + szR + 2 * szF + szV + szM))) return rc;
        char* base = (char*)ctx->ws;
        float* dI = (float*)base;
        float* dR = (float*)(base + szI);
        float* fb[2] = {(float*)(base + szI + szR), (float*)(base + szI + szR + szF)};
        double* dV = (double*)(base + szI + szR + 2 * szF);
        float* dM = (float*)(base + szI + szR + 2 * szF + szV);
        int cur = 0;
        for (int k = Lmax; k >= 0; k--) {
            int wk, hk, ks;
            double sg;
            nsof_farneback_level_size(64, 64, p.pyr_scale, k, &wk, &hk, &ks, &sg);   // blur taps depend on k only
            nsof_blur_taps btaps;
            if ((rc = nsof_host_blur_taps(ks, sg, &btaps)))
                return nsof_set_error(ctx, rc, "pyramid blur kernel size %d unsupported (max %d)", ks, NSOF_MAX_BLUR_TAPS - 1);
            const nsof_het_item* dt = d_tabs + (size_t)k * nh;
            const nsof_het_item* ht = tabs + (size_t)k * nh;
            const int nk_items = cnt[k];
            // incoming flow of the level (resample of the coarser level's field, zero for items that start here), level
            // image and expansion: grids over the largest extents of a size class, one launch per class
            for (const HetClass& c : classes[k]) {
                if ((rc = NSOF_PYR_SEL(ctx, nsof_launch_flow_upsample_het, c.count, dt + c.start, c.max_w, c.max_h, fb[cur],
                                       fb[cur ^ 1], (float)(1. / p.pyr_scale))))
                    return rc;
                // level 0: the expansion kernel forms the level image from the frames itself (see nsof_farneback_core)
                const bool u8 = k == 0 && btaps.ksize == 3 && !ctx->opt_pyr_fma && !ctx->opt_polyexp_f32;
                const float blur3[2] = {btaps.k[1], btaps.k[2]};
                if (!u8 && (rc = NSOF_PYR_SEL(ctx, nsof_launch_prep_het, c.count, dt + c.start, ht + c.start, k == 0, btaps, dI))) return rc;
                if ((rc = nsof_launch_polyexp_het(ctx, c.count, dt + c.start, c.max_w, c.max_h, ptaps, dI, dR, u8 ? blur3 : nullptr))) return rc;
            }
            cur ^= 1;
            for (int it = 0; it < p.iterations; it++) {
                const bool final = k == 0 && it == p.iterations - 1;
                if (exact_lat)
                    rc = nsof_launch_iterate_lat_het(ctx, nk_items, dt, max_w[k], max_h[k], dR, fb[cur], fb[cur ^ 1], final, p.winsize,
                                                     dM, dV);
                else if (exact_x)
                    rc = nsof_launch_iterate_x_het(ctx, nk_items, dt, max_w[k], max_h[k], dR, szR / 4, fb[cur], fb[cur ^ 1], final,
                                                   p.winsize, d_xj + xj_at[k], xj_stride[k], xj_jobs[k]);
                else if (exact)
                    rc = nsof_launch_iterate_het_exact(ctx, nk_items, dt, max_w[k], max_h[k], dR, fb[cur], fb[cur ^ 1], final,
                                                       p.winsize, dV);
                else
                    rc = nsof_launch_iterate_het(ctx, nk_items, dt, max_w[k], dR, fb[cur], fb[cur ^ 1], final, p.winsize);
                if (rc) return rc;
                cur ^= 1;
            }
        }
    }
    for (int i : rest)
        if (int rc = fallback_item(ctx, descs[i], p)) return rc;
    return NSOF_OK;
}

// ---- pipelined host entry ---------------------------------------------------------------------------------------
// CPUs of a NUMA node (sysfs cpulist "0-63,128-191"), empty if unknown.
std::vector<int> node_cpus(int node)
{
    std::vector<int> out;
    if (node < 0) return out;
    char path[96];
    snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
    FILE* f = fopen(path, "r");
    if (!f) return out;
    int a, b;
    char sep;
    while (fscanf(f, "%d", &a) == 1) {
        b = a;
        if (fscanf(f, "%c", &sep) == 1 && sep == '-') {
            if (fscanf(f, "%d", &b) != 1) b = a;
            if (fscanf(f, "%c", &sep) != 1) sep = 0;
        }
        for (int c = a; c <= b; c++) out.push_back(c);
        if (sep != ',') break;
    }
    fclose(f);
    return out;
}

thread_local int g_pack_node = -1;   // NUMA node the packing threads of THIS calling thread prefer (set per call from the context's device)

void parallel_rows(size_t n_tasks, const std::function<void(size_t)>& fn)
{
    static const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const unsigned nt = (unsigned)std::min<size_t>(std::min(16u, std::max(1u, hw / 2)), n_tasks);
    if (nt <= 1) {
        for (size_t i = 0; i < n_tasks; i++) fn(i);
        return;
    }
    static thread_local std::vector<int> cpus;
    static thread_local int cpus_node = -2;
    if (cpus_node != g_pack_node) {
        cpus = node_cpus(g_pack_node);
        cpus_node = g_pack_node;
    }
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; t++)
        th.emplace_back([&, t] {
            if (!cpus.empty()) {   // the staging buffers live on the GPU's node: copy from there (best effort)
                cpu_set_t set;
                CPU_ZERO(&set);
                for (int c : cpus)
                    if (c < CPU_SETSIZE) CPU_SET(c, &set);
                (void)sched_setaffinity(0, sizeof(set), &set);
            }
            for (size_t i = t; i < n_tasks; i += nt) fn(i);
        });
    for (auto& x : th) x.join();
}

}  // namespace

struct nsof_pipe {
    hipStream_t s_in = nullptr, s_out = nullptr;
    struct Slot {
        void* d_in = nullptr;  size_t d_in_bytes = 0;
        void* d_out = nullptr; size_t d_out_bytes = 0;
        void* h_in = nullptr;  size_t h_in_bytes = 0;
        void* h_out = nullptr; size_t h_out_bytes = 0;
        hipEvent_t in_done = nullptr, compute_done = nullptr, out_done = nullptr;
    } slot[3];
    static constexpr int NSLOT = 3;
};

void nsof_pipe_destroy(nsof_ctx* ctx)
{
    nsof_pipe* p = ctx->pipe;
    if (!p) return;
    for (auto& s : p->slot) {
        if (s.d_in) hipFree(s.d_in);
        if (s.d_out) hipFree(s.d_out);
        if (s.h_in) hipHostFree(s.h_in);
        if (s.h_out) hipHostFree(s.h_out);
        if (s.in_done) hipEventDestroy(s.in_done);
        if (s.compute_done) hipEventDestroy(s.compute_done);
        if (s.out_done) hipEventDestroy(s.out_done);
    }
    if (p->s_in) hipStreamDestroy(p->s_in);
    if (p->s_out) hipStreamDestroy(p->s_out);
    delete p;
    ctx->pipe = nullptr;
}

namespace {

int pipe_get(nsof_ctx* ctx, nsof_pipe** out)
{
    if (!ctx->pipe) {
        nsof_pipe* p = new (std::nothrow) nsof_pipe();
        if (!p) return nsof_set_error(ctx, NSOF_ENOMEM, "out of host memory");
        ctx->pipe = p;
        NSOF_HIP(ctx, hipStreamCreateWithFlags(&p->s_in, hipStreamNonBlocking));
        NSOF_HIP(ctx, hipStreamCreateWithFlags(&p->s_out, hipStreamNonBlocking));
        for (auto& s : p->slot) {
            NSOF_HIP(ctx, hipEventCreateWithFlags(&s.in_done, hipEventDisableTiming));
            NSOF_HIP(ctx, hipEventCreateWithFlags(&s.compute_done, hipEventDisableTiming));
            NSOF_HIP(ctx, hipEventCreateWithFlags(&s.out_done, hipEventDisableTiming));
        }
    }
    *out = ctx->pipe;
    return NSOF_OK;
}

int grow_dev(nsof_ctx* ctx, void** buf, size_t* cur, size_t need)
{
    if (*cur >= need) return NSOF_OK;
    if (*buf) NSOF_HIP(ctx, hipFree(*buf));
    *buf = nullptr; *cur = 0;
    if (hipMalloc(buf, need) != hipSuccess) { *buf = nullptr; return nsof_set_error(ctx, NSOF_ENOMEM, "hipMalloc(%zu)", need); }
    *cur = need;
    return NSOF_OK;
}
int grow_host(nsof_ctx* ctx, void** buf, size_t* cur, size_t need)
{
    if (*cur >= need) return NSOF_OK;
    if (*buf) NSOF_HIP(ctx, hipHostFree(*buf));
    *buf = nullptr; *cur = 0;
    *buf = nsof_pinned_alloc(ctx->device, need);
    if (!*buf) return nsof_set_error(ctx, NSOF_ENOMEM, "hipHostMalloc(%zu)", need);
    *cur = need;
    return NSOF_OK;
}

// Pinned (page-locked, HIP-registered) host memory can be the source / target of an asynchronous copy directly.
bool is_pinned(const void* p)
{
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();   // not a HIP allocation: clear the sticky error
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

struct Chunk {
    int lo, hi;                       // items [lo, hi)
    std::vector<size_t> in_off[2];    // byte offsets of the packed frames in the slot's input buffer
    std::vector<size_t> out_off;      // byte offsets of the dense flow fields in the slot's output buffer
    size_t in_bytes = 0, out_bytes = 0;
};

}  // namespace

extern "C" int nsof_farneback_u8_batch_desc_dev(nsof_ctx* ctx, int n_pairs, const nsof_pair_desc* pairs, double pyr_scale,
                                                int levels, int winsize, int iterations, int poly_n, double poly_sigma,
                                                int flags)
{
    if (!ctx) return NSOF_EINVAL;
    if (n_pairs < 0 || (n_pairs > 0 && !pairs)) return nsof_set_error(ctx, NSOF_EINVAL, "bad pair list");
    if (n_pairs > 32767) return nsof_set_error(ctx, NSOF_EUNSUPPORTED, "n_pairs=%d exceeds 32767 per call", n_pairs);
    if (n_pairs == 0) return NSOF_OK;
    const Params p{pyr_scale, levels, winsize, iterations, poly_n, poly_sigma, flags};
    return het_core(ctx, n_pairs, pairs, p);
}

// Ordered paste of the private crop fields of nsof_farneback_u8_roi_sequence_dev in ONE launch.  The reference's loop pastes
// a pair's crops one after the other, so where extended component boxes overlap the later component wins
// (optical_flow_seg.py:162).  A crop that overlaps an earlier one is computed into a private field; its pixels go to the
// canvas unless a LATER rectangle of the same gating frame covers them (that crop, private by construction, brings its own
// value): every canvas pixel then has exactly one writer among the pastes and the order is the reference's.  The sparse
// config-3 stream has 4425 such crops per 30 frames: as hipMemcpy2DAsync calls they were 12 ms of a 17 ms flow stage.
struct PasteRec {
    float* dst;                   // canvas position of the crop's first vector
    unsigned long long tmp_off;   // float offset of the private field in roi_tmp
    int x0, y0, w, h;
    int frame, idx;               // gating frame and the crop's index in its rectangle list
};
__global__ __launch_bounds__(256) void k_paste_ordered(const PasteRec* __restrict__ recs, const float* __restrict__ tmp,
                                                        const int32_t* __restrict__ counts, const int32_t* __restrict__ rects,
                                                        int max_rects, size_t canvas_pitch)
{
    __shared__ int4 later[256];
    __shared__ int n_later;
    const PasteRec r = recs[blockIdx.z];
    const int area = r.w * r.h;
    if ((int)(blockIdx.x * 1024) >= area) return;   // block-uniform
    const int cnt = min(counts[r.frame], max_rects);
    const int32_t* fr = rects + (size_t)r.frame * max_rects * 4;
    // the later rectangles that touch this crop at all (usually a handful)
    if (threadIdx.x == 0) n_later = 0;
    __syncthreads();
    for (int j = r.idx + 1 + (int)threadIdx.x; j < cnt; j += 256) {
        const int4 q = make_int4(fr[4 * j], fr[4 * j + 1], fr[4 * j + 2], fr[4 * j + 3]);
        if (q.z > q.x && q.w > q.y && q.x < r.x0 + r.w && r.x0 < q.z && q.y < r.y0 + r.h && r.y0 < q.w) {
            const int k = atomicAdd(&n_later, 1);
            if (k < 256) later[k] = q;
        }
    }
    __syncthreads();
    const int nl = min(n_later, 256);
    const float2* src = reinterpret_cast<const float2*>(tmp + r.tmp_off);
    float2* dst = reinterpret_cast<float2*>(r.dst);
    for (int i = blockIdx.x * 1024 + threadIdx.x; i < min(area, (int)(blockIdx.x + 1) * 1024); i += 256) {
        const int yy = i / r.w, xx = i - yy * r.w;
        const int X = r.x0 + xx, Y = r.y0 + yy;
        bool covered = false;
        for (int k = 0; k < nl && !covered; k++) covered = X >= later[k].x && X < later[k].z && Y >= later[k].y && Y < later[k].w;
        if (!covered) dst[(size_t)yy * canvas_pitch + xx] = src[i];
    }
}

// Gated sequence on the device: frames [n_frames][height] rows of row_stride bytes, the ROI table of the gating kernel
// (nsof_roi_from_surface_dev: counts [n_frames], rects [n_frames][max_rects][4] = x0, y0, x1, y1), flow canvases
// [n_frames - 1][height][width][2].  Pair k = (frame k, frame k + 1) is gated by the rectangles of frame k + gate_frame:
// gate_frame 1 = the map of the pair's SECOND frame, what opticalFlow3D is written to use (memimg2, optical_flow_seg.py:211-252);
// gate_frame 0 = the map of its FIRST frame, what the shipped scripts actually pass (memimg2 := memimg1, optical_flow_seg.py:435;
// the bug-compatible default of nsof.gating).  Every crop of every pair becomes one work item, results land in the zeroed
// canvases in place; a crop that overlaps an earlier crop of its pair (FLAG 1, extended component boxes) is computed into
// a private buffer and pasted afterwards, in label order, as the reference's loop overwrites.  The only traffic over
// PCIe is the rectangle table (16 bytes per ROI): the work list's shapes are needed on the host.
extern "C" int nsof_farneback_u8_roi_sequence_dev(nsof_ctx* ctx, int n_frames, const uint8_t* d_frames, ptrdiff_t row_stride,
                                                  ptrdiff_t frame_stride, int width, int height, const int32_t* d_counts,
                                                  const int32_t* d_rects, int max_rects, float* d_flows, double pyr_scale,
                                                  int levels, int winsize, int iterations, int poly_n, double poly_sigma,
                                                  int flags, int gate_frame, long long* n_calls, long long* n_pixels)
{
    if (!ctx) return NSOF_EINVAL;
    if (!d_frames || !d_counts || !d_rects || !d_flows || n_frames < 2 || max_rects < 1 || width < 1 || height < 1 ||
        row_stride < width || (gate_frame != 0 && gate_frame != 1))
        return nsof_set_error(ctx, NSOF_EINVAL, "roi_sequence: bad argument");
    NSOF_HIP(ctx, hipSetDevice(ctx->device));
    const size_t canvas = (size_t)width * height * 2;   // floats per pair
    NSOF_HIP(ctx, hipMemsetAsync(d_flows, 0, (size_t)(n_frames - 1) * canvas * 4, ctx->stream));
    std::vector<int32_t> counts(n_frames), rects((size_t)n_frames * max_rects * 4);
    NSOF_HIP(ctx, hipMemcpyAsync(counts.data(), d_counts, counts.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    NSOF_HIP(ctx, hipMemcpyAsync(rects.data(), d_rects, rects.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    NSOF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<nsof_pair_desc> descs;
    std::vector<PasteRec> pastes;
    std::vector<size_t> tmp_slot;            // index into descs of the items whose flow pointer is a roi_tmp offset
    size_t tmp_floats = 0;
    long long pixels = 0;
    for (int k = 0; k + 1 < n_frames; k++) {
        const int cnt = counts[k + gate_frame];
        if (cnt < 0 || cnt > max_rects)
            return nsof_set_error(ctx, NSOF_EINVAL, "roi_sequence: frame %d has %d rectangles, the table holds %d", k + gate_frame, cnt, max_rects);
        const int32_t* r = rects.data() + (size_t)(k + gate_frame) * max_rects * 4;
        for (int i = 0; i < cnt; i++) {
            const int x0 = r[4 * i], y0 = r[4 * i + 1], x1 = r[4 * i + 2], y1 = r[4 * i + 3];
            if (x1 <= x0 || y1 <= y0) continue;
            if (x0 < 0 || y0 < 0 || x1 > width || y1 > height)
                return nsof_set_error(ctx, NSOF_EINVAL, "roi_sequence: rectangle (%d,%d,%d,%d) leaves the %dx%d frame", x0, y0, x1, y1, width, height);
            bool overlap = false;
            for (int j = 0; j < i && !overlap; j++) {
                const int a0 = r[4 * j], b0 = r[4 * j + 1], a1 = r[4 * j + 2], b1 = r[4 * j + 3];
                overlap = a1 > a0 && b1 > b0 && x0 < a1 && a0 < x1 && y0 < b1 && b0 < y1;
            }
            nsof_pair_desc d;
            d.prev = d_frames + (size_t)k * frame_stride + (size_t)y0 * row_stride + x0;
            d.next = d_frames + (size_t)(k + 1) * frame_stride + (size_t)y0 * row_stride + x0;
            d.prev_stride = d.next_stride = row_stride;
            d.width = x1 - x0;
            d.height = y1 - y0;
            float* inplace = d_flows + (size_t)k * canvas + ((size_t)y0 * width + x0) * 2;
            if (overlap) {
                tmp_slot.push_back(descs.size());
                pastes.push_back(PasteRec{inplace, (unsigned long long)tmp_floats, x0, y0, d.width, d.height, k + gate_frame, i});
                d.flow = reinterpret_cast<float*>(tmp_floats * 4);     // offset for now: the buffer may still move
                d.flow_stride = (ptrdiff_t)d.width * 8;
                tmp_floats += (size_t)d.width * d.height * 2;
            } else {
                d.flow = inplace;
                d.flow_stride = (ptrdiff_t)width * 8;
            }
            pixels += (long long)d.width * d.height;
            descs.push_back(d);
        }
    }
    if (n_calls) *n_calls = (long long)descs.size();
    if (n_pixels) *n_pixels = pixels;
    if (descs.empty()) return NSOF_OK;
    if (tmp_floats) {
        if (int rc = nsof_ws_reserve(ctx, &ctx->roi_tmp, &ctx->roi_tmp_bytes, tmp_floats * 4)) return rc;
        for (size_t i : tmp_slot)
            descs[i].flow = reinterpret_cast<float*>((char*)ctx->roi_tmp + reinterpret_cast<uintptr_t>(descs[i].flow));
    }
    const Params p{pyr_scale, levels, winsize, iterations, poly_n, poly_sigma, flags};
    for (size_t i = 0; i < descs.size(); i += 32767) {
        const int n = (int)std::min<size_t>(32767, descs.size() - i);
        if (int rc = het_core(ctx, n, descs.data() + i, p)) return rc;
    }
    if (!pastes.empty()) {
        const size_t bytes = pastes.size() * sizeof(PasteRec);
        if (!ctx->paste_ev) NSOF_HIP(ctx, hipEventCreateWithFlags(&ctx->paste_ev, hipEventDisableTiming));
        else NSOF_HIP(ctx, hipEventSynchronize(ctx->paste_ev));   // the previous call's upload has left the pinned copy
        if (ctx->paste_bytes < bytes) {
            NSOF_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (ctx->paste_h) hipHostFree(ctx->paste_h);
            if (ctx->paste_d) hipFree(ctx->paste_d);
            ctx->paste_h = ctx->paste_d = nullptr;
            ctx->paste_bytes = 0;
            const size_t cap = align_up(bytes + bytes / 2, 4096);
            if (hipHostMalloc(&ctx->paste_h, cap, hipHostMallocDefault) != hipSuccess || hipMalloc(&ctx->paste_d, cap) != hipSuccess)
                return nsof_set_error(ctx, NSOF_ENOMEM, "paste table (%zu bytes)", cap);
            ctx->paste_bytes = cap;
        }
        memcpy(ctx->paste_h, pastes.data(), bytes);
        NSOF_HIP(ctx, hipMemcpyAsync(ctx->paste_d, ctx->paste_h, bytes, hipMemcpyHostToDevice, ctx->stream));
        NSOF_HIP(ctx, hipEventRecord(ctx->paste_ev, ctx->stream));
        int max_area = 0;
        for (const PasteRec& q : pastes) max_area = std::max(max_area, q.w * q.h);
        for (size_t i = 0; i < pastes.size(); i += 65535) {
            const unsigned n = (unsigned)std::min<size_t>(65535, pastes.size() - i);
            hipLaunchKernelGGL(k_paste_ordered, dim3((unsigned)((max_area + 1023) / 1024), 1, n), dim3(256), 0, ctx->stream,
                               (const PasteRec*)ctx->paste_d + i, (const float*)ctx->roi_tmp, d_counts, d_rects, max_rects,
                               (size_t)width);
        }
        NSOF_HIP(ctx, hipGetLastError());
    }
    return NSOF_OK;
}

extern "C" int nsof_farneback_u8_batch(nsof_ctx* ctx, int n_pairs, const nsof_pair_desc* pairs, double pyr_scale,
                                       int levels, int winsize, int iterations, int poly_n, double poly_sigma, int flags)
{
    if (!ctx) return NSOF_EINVAL;
    if (n_pairs < 0 || (n_pairs > 0 && !pairs)) return nsof_set_error(ctx, NSOF_EINVAL, "bad pair list");
    if (n_pairs == 0) return NSOF_OK;
    const Params p{pyr_scale, levels, winsize, iterations, poly_n, poly_sigma, flags};
    for (int i = 0; i < n_pairs; i++)
        if (int rc = validate_desc(ctx, i, pairs[i], p)) return rc;
    NSOF_HIP(ctx, hipSetDevice(ctx->device));
    nsof_pipe* pp;
    if (int rc = pipe_get(ctx, &pp)) return rc;
    g_pack_node = nsof_gpu_numa_node(ctx->device);

    // chunks of the list: about 512 MiB of flow (32 pairs of 1920x1080) each -- small enough that upload, compute
    // and download of neighbouring chunks overlap for lists of a hundred frames, large enough that the work list
    // of a chunk still fills the GPU (the download, not the compute, bounds the pipeline: 16.6 MB per 1080p pair)
    const char* chunk_env = getenv("NSOF_PIPE_CHUNK_MB");   // tests shrink it to force several chunks
    size_t budget = (size_t)std::max(1l, chunk_env ? atol(chunk_env) : 512l) << 20;
    if (!chunk_env) {
        // about 16 chunks per list, so that the pipeline's fill and drain (one upload + compute before the first
        // download, one download after the last compute) stay a small share: measured at 256 pairs of 1080p, 256 MiB
        // chunks 2.78 k pairs/s, 512 MiB 2.61 k, 128 MiB 2.15 k (8-pair work lists no longer fill the GPU)
        size_t total_out = 0;
        for (int i = 0; i < n_pairs; i++) total_out += (size_t)pairs[i].width * pairs[i].height * 8;
        budget = std::min<size_t>(512u << 20, std::max<size_t>(256u << 20, total_out / 16));
    }
    std::vector<Chunk> chunks;
    for (int i = 0; i < n_pairs;) {
        Chunk c;
        c.lo = i;
        while (i < n_pairs && (i == c.lo || (c.out_bytes < budget && i - c.lo < 8192))) {
            const size_t n0 = (size_t)pairs[i].width * pairs[i].height;
            for (int f = 0; f < 2; f++) { c.in_off[f].push_back(c.in_bytes); c.in_bytes += align_up(n0, 256); }
            c.out_off.push_back(c.out_bytes);
            c.out_bytes += align_up(n0 * 8, 256);
            i++;
        }
        c.hi = i;
        chunks.push_back(std::move(c));
    }
    std::vector<char> pin_in(2 * (size_t)n_pairs), pin_out(n_pairs);
    for (int i = 0; i < n_pairs; i++) {
        const nsof_pair_desc& d = pairs[i];
        pin_in[2 * i] = d.prev_stride == d.width && is_pinned(d.prev);
        pin_in[2 * i + 1] = d.next_stride == d.width && is_pinned(d.next);
        pin_out[i] = d.flow_stride == (ptrdiff_t)d.width * 8 && is_pinned(d.flow);
    }

    auto finish = [&](const Chunk& c, nsof_pipe::Slot& s) -> int {   // flows of a finished chunk -> caller's memory
        NSOF_HIP(ctx, hipEventSynchronize(s.out_done));
        std::vector<std::pair<int, int>> rows;   // (item, first row) tasks of 64 rows
        for (int i = c.lo; i < c.hi; i++)
            if (!pin_out[i])
                for (int y = 0; y < pairs[i].height; y += 64) rows.emplace_back(i, y);
        parallel_rows(rows.size(), [&](size_t t) {
            const int i = rows[t].first, y0 = rows[t].second;
            const nsof_pair_desc& d = pairs[i];
            const char* src = (const char*)s.h_out + c.out_off[i - c.lo];
            for (int y = y0; y < std::min(y0 + 64, d.height); y++)
                memcpy((char*)d.flow + (ptrdiff_t)y * d.flow_stride, src + (size_t)y * d.width * 8, (size_t)d.width * 8);
        });
        return NSOF_OK;
    };

    const int nc = (int)chunks.size();
    constexpr int NS = nsof_pipe::NSLOT;
    const char* fail_env = getenv("NSOF_PIPE_FAIL_AFTER_CHUNK");
    const int fail_after = fail_env ? atoi(fail_env) : -1;
    // NSOF_PIPE_TRACE=1: device-side begin/end of every stage of every chunk (timing events), printed at the end
    const bool trace = getenv("NSOF_PIPE_TRACE") != nullptr;
    std::vector<std::array<hipEvent_t, 6>> tev(trace ? nc : 0);
    std::vector<std::array<double, 2>> thost(trace ? nc : 0);
    auto now_ms = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_call = now_ms();
    auto mark = [&](int ci, int k, hipStream_t st) {
        if (!trace) return;
        hipEventCreate(&tev[ci][k]);
        hipEventRecord(tev[ci][k], st);
    };
    std::vector<nsof_pair_desc> dd;
    std::vector<char> finished(nc, 0);
    // Any early return below leaves copies in flight: uploads out of the caller's / the slots' pinned memory, downloads INTO
    // the caller's flow arrays.  The caller is free to release those buffers as soon as it sees the error, so every
    // non-OK exit drains the three streams first (and drops the trace events); the next call then finds idle slots.
    struct Drain {
        nsof_ctx* ctx; nsof_pipe* pp; std::vector<std::array<hipEvent_t, 6>>* tev; bool ok = false;
        ~Drain()
        {
            if (ok) return;
            (void)hipStreamSynchronize(pp->s_in);
            (void)hipStreamSynchronize(ctx->stream);
            (void)hipStreamSynchronize(pp->s_out);
            for (auto& a : *tev)
                for (auto e : a)
                    if (e) (void)hipEventDestroy(e);
        }
    } drain{ctx, pp, &tev};
    for (auto& a : tev) a.fill(nullptr);
    for (int ci = 0; ci < nc; ci++) {
        const Chunk& c = chunks[ci];
        nsof_pipe::Slot& s = pp->slot[ci % NS];
        const bool reuse = ci >= NS;           // the slot carried chunk ci - NS
        bool chunk_pageable_out = false;
        for (int i = c.lo; i < c.hi; i++) chunk_pageable_out = chunk_pageable_out || !pin_out[i];
        // Host-side waits only where host memory is reused: the slot's pinned input staging (its upload has long
        // finished) and, for pageable outputs, its pinned output staging (must be unpacked before the next download
        // lands in it).  Everything else is ordered on the GPU by events, so the host runs ahead and packs.
        if (reuse) {
            NSOF_HIP(ctx, hipEventSynchronize(s.in_done));
            if (!finished[ci - NS]) {
                bool prev_pageable = false;
                for (int i = chunks[ci - NS].lo; i < chunks[ci - NS].hi; i++) prev_pageable = prev_pageable || !pin_out[i];
                if (prev_pageable || s.h_out_bytes < c.out_bytes || s.d_out_bytes < c.out_bytes || s.d_in_bytes < c.in_bytes ||
                    s.h_in_bytes < c.in_bytes) {
                    if (int rc = finish(chunks[ci - NS], s)) return rc;
                    finished[ci - NS] = 1;
                }
            }
        }
        int rc;
        if ((rc = grow_dev(ctx, &s.d_in, &s.d_in_bytes, c.in_bytes)) || (rc = grow_dev(ctx, &s.d_out, &s.d_out_bytes, c.out_bytes)) ||
            (rc = grow_host(ctx, &s.h_in, &s.h_in_bytes, c.in_bytes)) ||
            (chunk_pageable_out && (rc = grow_host(ctx, &s.h_out, &s.h_out_bytes, c.out_bytes))))
            return rc;
        // stage 1: frames -> device (pageable / strided sources are packed into the pinned slot buffer by a few threads)
        if (trace) thost[ci][0] = now_ms() - t_call;
        std::vector<std::array<int, 3>> rows;   // (item, frame, first row)
        for (int i = c.lo; i < c.hi; i++)
            for (int f = 0; f < 2; f++)
                if (!pin_in[2 * i + f])
                    for (int y = 0; y < pairs[i].height; y += 128) rows.push_back({i, f, y});
        parallel_rows(rows.size(), [&](size_t t) {
            const int i = rows[t][0], f = rows[t][1], y0 = rows[t][2];
            const nsof_pair_desc& d = pairs[i];
            const uint8_t* src = f ? d.next : d.prev;
            const ptrdiff_t st = f ? d.next_stride : d.prev_stride;
            uint8_t* dst = (uint8_t*)s.h_in + c.in_off[f][i - c.lo];
            const int y1 = std::min(y0 + 128, d.height);
            if (st == (ptrdiff_t)d.width) {
                memcpy(dst + (size_t)y0 * d.width, src + (ptrdiff_t)y0 * st, (size_t)(y1 - y0) * d.width);
            } else {
                for (int y = y0; y < y1; y++) memcpy(dst + (size_t)y * d.width, src + (ptrdiff_t)y * st, (size_t)d.width);
            }
        });
        if (trace) thost[ci][1] = now_ms() - t_call;
        if (reuse) NSOF_HIP(ctx, hipStreamWaitEvent(pp->s_in, s.compute_done, 0));   // d_in was read by chunk ci - NS
        mark(ci, 0, pp->s_in);
        bool all_packed = true;
        for (int i = c.lo; i < c.hi; i++) all_packed = all_packed && !pin_in[2 * i] && !pin_in[2 * i + 1];
        if (all_packed) {
            NSOF_HIP(ctx, hipMemcpyAsync(s.d_in, s.h_in, c.in_bytes, hipMemcpyHostToDevice, pp->s_in));
        } else {
            for (int i = c.lo; i < c.hi; i++)
                for (int f = 0; f < 2; f++) {
                    const size_t off = c.in_off[f][i - c.lo], n0 = (size_t)pairs[i].width * pairs[i].height;
                    const void* src = pin_in[2 * i + f] ? (const void*)(f ? pairs[i].next : pairs[i].prev)
                                                        : (const void*)((char*)s.h_in + off);
                    NSOF_HIP(ctx, hipMemcpyAsync((char*)s.d_in + off, src, n0, hipMemcpyHostToDevice, pp->s_in));
                }
        }
        NSOF_HIP(ctx, hipEventRecord(s.in_done, pp->s_in));
        mark(ci, 1, pp->s_in);
        // stage 2: compute on the context's stream
        NSOF_HIP(ctx, hipStreamWaitEvent(ctx->stream, s.in_done, 0));
        if (reuse) NSOF_HIP(ctx, hipStreamWaitEvent(ctx->stream, s.out_done, 0));   // d_out is being downloaded (chunk ci - NS)
        dd.resize(c.hi - c.lo);
        for (int i = c.lo; i < c.hi; i++) {
            nsof_pair_desc& d = dd[i - c.lo];
            d.prev = (const uint8_t*)s.d_in + c.in_off[0][i - c.lo];
            d.next = (const uint8_t*)s.d_in + c.in_off[1][i - c.lo];
            d.prev_stride = d.next_stride = pairs[i].width;
            d.width = pairs[i].width;
            d.height = pairs[i].height;
            d.flow = (float*)((char*)s.d_out + c.out_off[i - c.lo]);
            d.flow_stride = (ptrdiff_t)pairs[i].width * 8;
        }
        mark(ci, 2, ctx->stream);
        if ((rc = het_core(ctx, c.hi - c.lo, dd.data(), p))) return rc;
        if (fail_after >= 0 && ci == fail_after)   // NSOF_PIPE_FAIL_AFTER_CHUNK (tests): an error with copies in flight
            return nsof_set_error(ctx, NSOF_ENOMEM, "NSOF_PIPE_FAIL_AFTER_CHUNK=%d: injected failure", fail_after);
        NSOF_HIP(ctx, hipEventRecord(s.compute_done, ctx->stream));
        mark(ci, 3, ctx->stream);
        // stage 3: flow -> host
        NSOF_HIP(ctx, hipStreamWaitEvent(pp->s_out, s.compute_done, 0));
        mark(ci, 4, pp->s_out);
        if (!chunk_pageable_out) {
            for (int i = c.lo; i < c.hi; i++) {
                const size_t off = c.out_off[i - c.lo], nb = (size_t)pairs[i].width * pairs[i].height * 8;
                NSOF_HIP(ctx, hipMemcpyAsync(pairs[i].flow, (char*)s.d_out + off, nb, hipMemcpyDeviceToHost, pp->s_out));
            }
        } else {
            bool none_pinned = true;
            for (int i = c.lo; i < c.hi; i++) none_pinned = none_pinned && !pin_out[i];
            if (none_pinned) {
                NSOF_HIP(ctx, hipMemcpyAsync(s.h_out, s.d_out, c.out_bytes, hipMemcpyDeviceToHost, pp->s_out));
            } else {
                for (int i = c.lo; i < c.hi; i++) {
                    const size_t off = c.out_off[i - c.lo], nb = (size_t)pairs[i].width * pairs[i].height * 8;
                    void* dst = pin_out[i] ? (void*)pairs[i].flow : (void*)((char*)s.h_out + off);
                    NSOF_HIP(ctx, hipMemcpyAsync(dst, (char*)s.d_out + off, nb, hipMemcpyDeviceToHost, pp->s_out));
                }
            }
        }
        NSOF_HIP(ctx, hipEventRecord(s.out_done, pp->s_out));
        mark(ci, 5, pp->s_out);
    }
    for (int ci = 0; ci < nc; ci++)
        if (!finished[ci])
            if (int rc = finish(chunks[ci], pp->slot[ci % NS])) return rc;
    if (int rcs = nsof_stream_sync_checked(ctx)) return rcs;   // incl. a lost hand-over of the exact-order flow kernels
    if (trace) {
        fprintf(stderr, "[nsof pipe] %d chunks, host total %.2f ms; per chunk: pairs | host pack begin..end | h2d | compute | d2h (ms from first upload)\n",
                nc, now_ms() - t_call);
        for (int ci = 0; ci < nc; ci++) {
            float t[6];
            for (int k = 0; k < 6; k++) hipEventElapsedTime(&t[k], tev[0][0], tev[ci][k]);
            fprintf(stderr, "[nsof pipe] %3d: %4d | %7.2f..%7.2f | %7.2f..%7.2f | %7.2f..%7.2f | %7.2f..%7.2f\n", ci,
                    chunks[ci].hi - chunks[ci].lo, thost[ci][0], thost[ci][1], t[0], t[1], t[2], t[3], t[4], t[5]);
        }
        for (auto& a : tev)
            for (auto e : a) hipEventDestroy(e);
    }
    drain.ok = true;
    return NSOF_OK;
}

extern "C" void* nsof_host_alloc(size_t bytes)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        (void)hipGetLastError();
        dev = 0;
    }
    return nsof_pinned_alloc(dev, bytes);   // on the current device's NUMA node where the host lets us choose
}

extern "C" void nsof_host_free(void* p)
{
    if (p) (void)hipHostFree(p);
}
