// splatco_amd/csrc/capi.hip -- the C-ABI of include/splatco_raster.h over the gfx950 kernels.
// Every data buffer is caller-owned (several forward graphs are alive at once in the mv loop of
// train.py:171-240).  Process-lifetime state of the library, all of it below: a thread-local error string, one pinned
// 64-byte mailbox per calling host thread + a process-wide stamp counter for the plan read-backs, a per-device flag for
// the dynamic-LDS attribute, and the opt-in profiling event pool (g_prof_*; single-threaded, see the header).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "common.h"
#include "adam.h"
#include "tv.h"
#include <chrono>
#include <cstring>

using namespace scr;

static thread_local char g_err[512] = "";
namespace scr { int g_force_deep_lists = -1; }

static int fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return 1;
}

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) return fail("%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// after a kernel launch: always catch launch errors; in debug mode also run it to completion
#define CHECK_LAUNCH(name, debug, st)                                                      \
    do {                                                                                   \
        hipError_t e_ = hipGetLastError();                                                 \
        if (e_ != hipSuccess) return fail("launch of %s failed: %s", name, hipGetErrorString(e_)); \
        if (debug) {                                                                       \
            e_ = hipStreamSynchronize(st);                                                 \
            if (e_ != hipSuccess) return fail("%s faulted: %s", name, hipGetErrorString(e_)); \
        }                                                                                  \
    } while (0)

// ---- opt-in kernel timing (scr_profile_*): hipEvent pairs on the launch stream
#include <dlfcn.h>
#include <atomic>
#include <vector>
namespace {
struct ProfRec { int idx; hipEvent_t a, b; };
unsigned g_prof_mask = 0;  // bit i set: kernel class i is timed
unsigned g_prof_stride = 1;             // every g_prof_stride-th launch of a timed class is bracketed (scr_profile_stride)
unsigned g_prof_seen[32] = {};          // launches of class i since the mask was set
std::vector<ProfRec> g_prof_recs;
std::vector<hipEvent_t> g_prof_pool;
hipEvent_t prof_event() {
    hipEvent_t e;
    if (!g_prof_pool.empty()) { e = g_prof_pool.back(); g_prof_pool.pop_back(); return e; }
    (void)hipEventCreate(&e);
    return e;
}
// ---- opt-in stage markers (scr_markers_enable): roctx ranges on the calling thread, the library loaded on first use
bool g_mark_on = false;
int (*g_roctx_push)(const char*) = nullptr;
int (*g_roctx_pop)() = nullptr;
struct MarkScope {
    bool on;
    explicit MarkScope(const char* name) : on(g_mark_on) { if (on) (void)g_roctx_push(name); }
    ~MarkScope() { if (on) (void)g_roctx_pop(); }
};
#define SCR_MARK_FN MarkScope mark_fn_(__func__)
extern const char* const kProfNames[SCR_PROF_COUNT];
struct ProfScope {
    bool on; hipStream_t st; ProfRec r; MarkScope mark;
    ProfScope(int idx, hipStream_t s) : on((g_prof_mask >> idx) & 1u), st(s), mark(kProfNames[idx]) {
        if (on && g_prof_stride > 1) on = (g_prof_seen[idx]++ % g_prof_stride) == 0;
        if (on) { r.idx = idx; r.a = prof_event(); r.b = prof_event(); (void)hipEventRecord(r.a, st); }
    }
    ~ProfScope() { if (on) { (void)hipEventRecord(r.b, st); g_prof_recs.push_back(r); } }
};
const char* const kProfNames[SCR_PROF_COUNT] = {
    "filter_kernel", "preprocess_kernel", "plan_scan_kernel", "scatter_kernel", "tile_sort_kernel",
    "blend_forward_kernel", "blend_backward_kernel", "preprocess_backward_kernel", "expand_kernel",
    "expand_backward_kernel", "plane_sample_backward_kernels", "l1_ssim_forward_kernel",
    "l1_ssim_backward_kernel", "triplane_forward_kernel", "mlp_heads_kernel", "mlp_heads_backward_kernel",
    "norm_linear_kernels", "norm_linear_backward_kernels", "plane_attention_kernels"};
}  // namespace

// streaming copy, 16 B per lane, four loads in flight per thread, non-temporal: the shape that reaches the highest HBM
// bandwidth on MI355X among those of tools/exp/copy_probe.hip (6.3 TB/s read + write; a grid-stride loop with few
// workgroups stays at 4.7).  bench.py quotes its rate as the achievable peak next to the 8 TB/s datasheet figure.
typedef float copy_f4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) copy_probe_kernel(const copy_f4* __restrict__ src, copy_f4* __restrict__ dst, size_t n) {
    const size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x;
    copy_f4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) if (base + u * 256 < n) v[u] = __builtin_nontemporal_load(&src[base + u * 256]);
#pragma unroll
    for (int u = 0; u < 4; ++u) if (base + u * 256 < n) __builtin_nontemporal_store(v[u], &dst[base + u * 256]);
}

// pinned host memory the GPU writes and the host polls (scr_forward_plan); per host thread, lives for the process
struct Mailbox {
    volatile unsigned long long* host = nullptr;
    unsigned long long* dev = nullptr;
    unsigned long long seq = 0;
};
static Mailbox& mailbox() {
    thread_local Mailbox mb;
    thread_local bool tried = false;
    if (!tried) {
        tried = true;
        void* h = nullptr;
        if (hipHostMalloc(&h, 64, hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess && h) {
            void* d = nullptr;
            if (hipHostGetDevicePointer(&d, h, 0) == hipSuccess && d) {
                memset(h, 0, 64);
                mb.host = (volatile unsigned long long*)h;
                mb.dev = (unsigned long long*)d;
            } else {
                (void)hipHostFree(h);
            }
        }
        (void)hipGetLastError();
    }
    return mb;
}

// spin until the kernel(s) of this call have posted stamp `seq` (slot 1, and slot 3 when `two`); false when the
// stream finished without a visible post (caller falls back to a copy + stream synchronisation)
static bool mailbox_wait(Mailbox& mb, unsigned long long seq, bool two, hipStream_t st) {
    if (!mb.host) return false;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; ++spins) {
        // stamp first, with acquire ordering: the value words read afterwards are at least as new as the stamp
        if (__atomic_load_n(&mb.host[1], __ATOMIC_ACQUIRE) == seq && (!two || __atomic_load_n(&mb.host[3], __ATOMIC_ACQUIRE) == seq))
            return true;
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
        if ((spins & 1023u) == 1023u) {
            if (hipStreamQuery(st) != hipErrorNotReady)
                return __atomic_load_n(&mb.host[1], __ATOMIC_ACQUIRE) == seq &&
                       (!two || __atomic_load_n(&mb.host[3], __ATOMIC_ACQUIRE) == seq);
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20)) return false;
        }
    }
}

static int check_settings(const scr_settings* s) {
    if (!s) return fail("settings is NULL");
    if (s->image_height <= 0 || s->image_width <= 0) return fail("image size must be positive");
    if (s->image_width > 16 * 65535 || s->image_height > 16 * 65535) return fail("image too large for 16-bit tile coordinates");
    if (!s->bg || !s->viewmatrix || !s->projmatrix || !s->campos) return fail("bg / viewmatrix / projmatrix / campos must be device pointers");
    if (s->sh_degree < 0 || s->sh_degree > 3) return fail("sh_degree must be 0..3");
    return 0;
}

extern "C" {

int scr_abi_version(void) { return SCR_ABI_VERSION; }
const char* scr_last_error(void) { return g_err; }

size_t scr_geom_bytes(int64_t P, int32_t H, int32_t W) { return geom_view(nullptr, P, H, W).bytes; }
size_t scr_binning_bytes(int64_t I, int64_t max_tile) { return bin_view(nullptr, I, max_tile).bytes; }
size_t scr_image_bytes(int32_t H, int32_t W) { return img_view(nullptr, H, W).bytes; }
size_t scr_backward_scratch_bytes(int64_t I) { return align_up((size_t)(I > 0 ? I : 1) * sizeof(GradRec)); }

int scr_visible_filter(int64_t P, const float* means3D, const float* scales, const float* rotations,
                       const float* cov3D_precomp, const scr_settings* settings, int32_t* radii_out,
                       void* stream) {
    SCR_MARK_FN;
    if (check_settings(settings)) return 1;
    if (P < 0) return fail("P < 0");
    if (P == 0) return 0;
    if (!means3D || !radii_out) return fail("means3D / radii_out is NULL");
    if (!cov3D_precomp && !(scales && rotations)) return fail("provide (scales, rotations) or cov3D_precomp");
    hipStream_t st = (hipStream_t)stream;
    { ProfScope ps_(SCR_PROF_FILTER, st);
      launch_filter(P, means3D, scales, rotations, cov3D_precomp, ksettings(settings), radii_out, st); }
    CHECK_LAUNCH("filter_kernel", settings->debug, st);
    return 0;
}

int scr_mark_visible(int64_t P, const float* means3D, const float* viewmatrix, uint8_t* present_out,
                     void* stream) {
    SCR_MARK_FN;
    if (P < 0) return fail("P < 0");
    if (P == 0) return 0;
    if (!means3D || !viewmatrix || !present_out) return fail("NULL argument");
    hipStream_t st = (hipStream_t)stream;
    launch_mark_visible(P, means3D, viewmatrix, present_out, st);
    CHECK_LAUNCH("mark_visible_kernel", 0, st);
    return 0;
}

// phase 1, first half: argument checks, tile counts, projection, scans; the two counts are on their way to the mailbox
static int plan_enqueue(int64_t P, int32_t M, const float* means3D, const float* scales, const float* rotations,
                        const float* cov3D_precomp, const float* opacities, const float* shs, const float* colors_precomp,
                        const scr_settings* settings, void* geom_buf, int32_t* radii_out, int64_t* plan_host, hipStream_t st,
                        unsigned long long& seq_out) {
    if (check_settings(settings)) return 1;
    if (P < 0) return fail("P < 0");
    if (!plan_host) return fail("plan_host is NULL");
    plan_host[0] = plan_host[1] = plan_host[3] = 0;
    if (!geom_buf) return fail("geom_buf is NULL");
    if ((shs != nullptr) == (colors_precomp != nullptr))
        return fail("pass either shs or colors_precomp (one of them, not both, not neither)");
    if (((scales != nullptr) || (rotations != nullptr)) == (cov3D_precomp != nullptr) || ((scales != nullptr) != (rotations != nullptr)))
        if (!(cov3D_precomp && !scales && !rotations) && !(scales && rotations && !cov3D_precomp))
            return fail("pass either scales + rotations or cov3D_precomp (one form of the covariance)");
    if (P > 0 && (!means3D || !opacities || !radii_out)) return fail("means3D / opacities / radii_out is NULL");
    if (P >= (1ll << ID_BITS)) return fail("P = %lld exceeds the 2^%d Gaussians a sort key can index", (long long)P, ID_BITS);
    if (shs && (M < (settings->sh_degree + 1) * (settings->sh_degree + 1)))
        return fail("shs has %d coefficients, sh_degree %d needs %d", M, settings->sh_degree,
                    (settings->sh_degree + 1) * (settings->sh_degree + 1));
    KSettings ks = ksettings(settings);
    GeomView gv = geom_view(geom_buf, P, ks.H, ks.W);
    Grid g(ks.H, ks.W);
    { ZeroList z; z.add(gv.tile_count, (size_t)g.tiles * 4, st); z.add(gv.total + 3, 8, st); launch_zero(z, st); }
    { ProfScope ps_(SCR_PROF_PREPROCESS, st);
      launch_preprocess(P, M, means3D, scales, rotations, cov3D_precomp, opacities, shs, colors_precomp, ks, gv,
                        radii_out, st); }
    CHECK_LAUNCH("preprocess_kernel", settings->debug, st);
    // The two counts come back through a small pinned, device-visible mailbox (one per host thread, created on
    // first use): the scan kernel posts them with a sequence stamp and this thread spins on the stamp.  A blocking
    // hipStreamSynchronize + 16-byte copy costs a copy launch and, worse, a wake-up of the sleeping thread, which
    // on a busy host is anywhere between 10 and 200 us of idle GPU in the middle of every forward pass.
    Mailbox& mb = mailbox();
    seq_out = ++mb.seq;
    { ProfScope ps_(SCR_PROF_PLAN_SCAN, st); launch_plan_scans(P, ks, gv, mb.dev, seq_out, st); }
    CHECK_LAUNCH("plan_scan_kernel", settings->debug, st);
    return 0;
}

// phase 1, second half: wait for the two counts
static int plan_wait(const scr_settings* settings, void* geom_buf, int64_t P, unsigned long long seq, int64_t* plan_host,
                     hipStream_t st) {
    KSettings ks = ksettings(settings);
    GeomView gv = geom_view(geom_buf, P, ks.H, ks.W);
    Mailbox& mb = mailbox();
    unsigned long long total[4] = {0, 0, 0, 0};      // instances, largest tile, -, plan flags
    const bool posted = mailbox_wait(mb, seq, true, st);
    if (posted) {
        total[0] = mb.host[0];
        total[1] = mb.host[2] & 0xffffffffull;
        total[3] = mb.host[2] >> 32;
    }
    if (!posted) {  // no mailbox, or its writes are not visible on this system: the classic read-back
        HIP_TRY(hipMemcpyAsync(total, gv.total, 32, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    // per-workgroup sums saturate at 2^32 - 1 (preprocess_kernel) and the scan adds them in 64 bits: any total at or above
    // 2^32 - 1 means "does not fit" (the prefix values are meaningless then; nothing has been written through them yet)
    if (total[0] >= 0xFFFFFFFFull)
        return fail("num_rendered >= 2^32 - 1 (counted %llu): the (Gaussian, tile) instances do not fit 32-bit indices", total[0]);
    plan_host[0] = (int64_t)total[0];
    plan_host[1] = (int64_t)total[1];
    plan_host[3] = (int64_t)total[3];
    return 0;
}

int scr_forward_plan(int64_t P, int32_t M, const float* means3D, const float* scales,
                     const float* rotations, const float* cov3D_precomp, const float* opacities,
                     const float* shs, const float* colors_precomp, const scr_settings* settings,
                     void* geom_buf, int32_t* radii_out, int64_t* plan_host, void* stream) {
    SCR_MARK_FN;
    unsigned long long seq = 0;
    const int rc = plan_enqueue(P, M, means3D, scales, rotations, cov3D_precomp, opacities, shs, colors_precomp, settings, geom_buf,
                                radii_out, plan_host, (hipStream_t)stream, seq);
    return rc ? rc : plan_wait(settings, geom_buf, P, seq, plan_host, (hipStream_t)stream);
}

static int forward_run_impl(int64_t P, int64_t I, int64_t max_tile, int64_t plan_flags, const scr_settings* settings, void* geom_buf,
                            void* binning_buf, void* image_buf, float* out_color, void* stream, bool scatter_done);

int scr_forward_run(int64_t P, int64_t I, int64_t max_tile, int64_t plan_flags, const scr_settings* settings, void* geom_buf,
                    void* binning_buf, void* image_buf, float* out_color, void* stream) {
    SCR_MARK_FN;
    return forward_run_impl(P, I, max_tile, plan_flags, settings, geom_buf, binning_buf, image_buf, out_color, stream, false);
}

static int forward_run_impl(int64_t P, int64_t I, int64_t max_tile, int64_t plan_flags, const scr_settings* settings, void* geom_buf,
                            void* binning_buf, void* image_buf, float* out_color, void* stream, bool scatter_done) {
    if (check_settings(settings)) return 1;
    if (plan_flags & ~(int64_t)(SCR_PLAN_NONFINITE_COLOUR | SCR_PLAN_LARGE_RECTS)) return fail("plan_flags %lld: not a value scr_forward_plan returned", (long long)plan_flags);
    if (!geom_buf || !binning_buf || !image_buf || !out_color) return fail("NULL buffer");
    hipStream_t st = (hipStream_t)stream;
    KSettings ks = ksettings(settings);
    GeomView gv = geom_view(geom_buf, P, ks.H, ks.W);
    BinView bv = bin_view(binning_buf, I, max_tile);
    ImgView iv = img_view(image_buf, ks.H, ks.W);
    if (I > 0) {
        if (!scatter_done) {
            { ProfScope ps_(SCR_PROF_SCATTER, st); launch_scatter(P, ks, gv, bv, ~0ull, st); }
            CHECK_LAUNCH("scatter_kernel", settings->debug, st);
        }
        { ProfScope ps_(SCR_PROF_TILE_SORT, st); launch_tile_sort(ks, gv, bv, max_tile, !deep_lists(I, Grid(ks.H, ks.W).tiles), st); }
        CHECK_LAUNCH("tile_sort_kernel", settings->debug, st);
    }
    { ProfScope ps_(SCR_PROF_BLEND_FORWARD, st); launch_blend_forward(ks, gv, bv, iv, out_color, 2 * max_tile * (int64_t)Grid(ks.H, ks.W).tiles > 3 * I,
                                                                      (plan_flags & SCR_PLAN_NONFINITE_COLOUR) != 0, st); }
    CHECK_LAUNCH("blend_forward_kernel", settings->debug, st);
    return 0;
}

int scr_forward_plan_run(int64_t P, int32_t M, const float* means3D, const float* scales, const float* rotations,
                         const float* cov3D_precomp, const float* opacities, const float* shs, const float* colors_precomp,
                         const scr_settings* settings, void* geom_buf, int32_t* radii_out, int64_t* plan_host,
                         void* binning_buf, size_t binning_capacity_bytes, void* image_buf, float* out_color, void* stream) {
    SCR_MARK_FN;
    if (!plan_host) return fail("plan_host is NULL");
    plan_host[2] = 0;
    hipStream_t st = (hipStream_t)stream;
    unsigned long long seq = 0;
    int rc = plan_enqueue(P, M, means3D, scales, rotations, cov3D_precomp, opacities, shs, colors_precomp, settings, geom_buf,
                          radii_out, plan_host, st, seq);
    if (rc) return rc;
    // The scatter kernel goes out BEFORE this thread has seen the instance count: it needs nothing the host knows (the
    // keys are the first array of the binning buffer whatever the count), only room -- and checks on the device that the
    // count fits (17 bytes per instance is the least a binning buffer of that many instances takes, so the keys fit).
    // While it runs, the count arrives, the sort and the blend are queued behind it: the 12 - 20 us the GPU used to idle
    // in the middle of every forward pass (profiles/r03x_step_gaps.txt) are gone.
    const unsigned long long cap = binning_buf ? (unsigned long long)(binning_capacity_bytes / 17) : 0ull;
    const bool early = binning_buf && P > 0 && cap > 0 && image_buf && out_color;
    if (early) {
        KSettings ks = ksettings(settings);
        GeomView gv = geom_view(geom_buf, P, ks.H, ks.W);
        BinView bv = bin_view(binning_buf, 0, 0);
        { ProfScope ps_(SCR_PROF_SCATTER, st); launch_scatter(P, ks, gv, bv, cap, st); }
        CHECK_LAUNCH("scatter_kernel", settings->debug, st);
    }
    rc = plan_wait(settings, geom_buf, P, seq, plan_host, st);
    if (rc) return rc;
    const bool fits = binning_buf && scr_binning_bytes(plan_host[0], plan_host[1]) <= binning_capacity_bytes;
    const bool scattered = early && (unsigned long long)plan_host[0] <= cap;
    if (!fits) {
        if (scattered && plan_host[0] > 0) {      // the early scatter ran and used up the cursors: give scr_forward_run fresh ones
            KSettings ks = ksettings(settings);
            GeomView gv = geom_view(geom_buf, P, ks.H, ks.W);
            ZeroList z;
            z.add(gv.cursor, (size_t)Grid(ks.H, ks.W).tiles * 4, st);
            launch_zero(z, st);
        }
        return 0;                                  // caller allocates, then scr_forward_run
    }
    rc = forward_run_impl(P, plan_host[0], plan_host[1], plan_host[3], settings, geom_buf, binning_buf, image_buf, out_color, stream, scattered);
    if (rc) return rc;
    plan_host[2] = 1;
    return 0;
}

int scr_backward(int64_t P, int32_t M, int64_t I, int64_t plan_flags, const float* means3D, const float* scales,
                 const float* rotations, const float* cov3D_precomp, const float* shs,
                 const scr_settings* settings, const int32_t* radii, void* geom_buf,
                 const void* binning_buf, void* image_buf, const float* dL_dcolor, void* scratch,
                 float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dcolors, float* dL_dsh,
                 float* dL_dopacity, float* dL_dscales, float* dL_drotations, float* dL_dcov3D,
                 void* stream) {
    SCR_MARK_FN;
    if (check_settings(settings)) return 1;
    if (P == 0) return 0;
    if (plan_flags & ~(int64_t)(SCR_PLAN_NONFINITE_COLOUR | SCR_PLAN_LARGE_RECTS)) return fail("plan_flags %lld: not a value scr_forward_plan returned", (long long)plan_flags);
    if (!geom_buf || !binning_buf || !image_buf || !dL_dcolor || !scratch) return fail("NULL buffer");
    if (!means3D || !radii || !dL_dmeans3D || !dL_dmeans2D || !dL_dopacity) return fail("NULL argument");
    if (shs ? !dL_dsh : !dL_dcolors) return fail("colour gradient output missing");
    if (cov3D_precomp ? !dL_dcov3D : !(dL_dscales && dL_drotations && scales && rotations))
        return fail("covariance gradient output missing");
    hipStream_t st = (hipStream_t)stream;
    KSettings ks = ksettings(settings);
    GeomView gv = geom_view(geom_buf, P, ks.H, ks.W);
    BinView bv = bin_view((void*)binning_buf, I, 0);  // the lists read here come first in the layout
    ImgView iv = img_view(image_buf, ks.H, ks.W);
    // a value no earlier call of this process used (and that uninitialised memory is unlikely to hold): see blend.hip
    static std::atomic<unsigned long long> stamp_counter{0x5ca1ab1e00000000ull};
    const unsigned long long stamp = ++stamp_counter;
    if (settings->debug) {      // the flags the caller carried from scr_forward_plan against the ones the forward left in geom_buf
        unsigned long long dev_flags = 0;
        HIP_TRY(hipMemcpyAsync(&dev_flags, gv.total + 3, 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if ((long long)dev_flags != (long long)plan_flags)
            return fail("plan_flags %lld passed to scr_backward, but the forward pass of this geom_buf raised %llu", (long long)plan_flags, dev_flags);
    }
    if (I > 0) {
        // records 32.. of large rects are cleared whatever plan_flags says: the kernel reads the forward's verdict from
        // geom_buf and leaves at once when there are none (2 us), so a stale argument cannot make preprocess_backward sum
        // uninitialised scratch
        launch_zero_far_records(P, gv, (GradRec*)scratch, st);
        CHECK_LAUNCH("zero_far_records_kernel", settings->debug, st);
        { ProfScope ps_(SCR_PROF_BLEND_BACKWARD, st);
          launch_blend_backward(ks, gv, bv, iv, dL_dcolor, (GradRec*)scratch, stamp, deep_lists(I, Grid(ks.H, ks.W).tiles),
                                record_flags(I, Grid(ks.H, ks.W).tiles), (plan_flags & SCR_PLAN_NONFINITE_COLOUR) != 0, st); }
        CHECK_LAUNCH("blend_backward_kernel", settings->debug, st);
    }
    { ProfScope ps_(SCR_PROF_PREPROCESS_BACKWARD, st);
      launch_preprocess_backward(P, M, means3D, scales, rotations, cov3D_precomp, shs, ks, radii, gv, bv,
                                 (const GradRec*)scratch, iv.cut_key, stamp, record_flags(I, Grid(ks.H, ks.W).tiles), dL_dmeans3D, dL_dmeans2D, shs ? nullptr : dL_dcolors,
                                 shs ? dL_dsh : nullptr, dL_dopacity, cov3D_precomp ? nullptr : dL_dscales,
                                 cov3D_precomp ? nullptr : dL_drotations, cov3D_precomp ? dL_dcov3D : nullptr, st); }
    CHECK_LAUNCH("preprocess_backward_kernel", settings->debug, st);
    return 0;
}

int scr_debug_force_deep_lists(int mode) {
    if (mode < -1 || mode > 1) return fail("scr_debug_force_deep_lists: -1 (automatic), 0 or 1");
    scr::g_force_deep_lists = mode;
    return 0;
}

int scr_debug_get(int which, int64_t P, int64_t I, int32_t H, int32_t W, const void* geom_buf,
                  const void* binning_buf, const void* image_buf, void* out, void* stream) {
    SCR_MARK_FN;
    hipStream_t st = (hipStream_t)stream;
    Grid g(H, W);
    const void* src = nullptr;
    size_t bytes = 0;
    if (which <= SCR_DBG_RANGES || which == SCR_DBG_SPLAT_RECORDS) {
        if (!geom_buf) return fail("geom_buf is NULL");
        GeomView gv = geom_view((void*)geom_buf, P, H, W);
        if (which == SCR_DBG_TILES_TOUCHED) { src = gv.tiles_touched; bytes = (size_t)P * 4; }
        else if (which == SCR_DBG_POINT_OFFSETS) { src = gv.point_offsets; bytes = (size_t)P * 4; }
        else if (which == SCR_DBG_RANGES) { src = gv.ranges; bytes = (size_t)g.tiles * 8; }
        else { src = gv.rec; bytes = (size_t)P * REC_F * 4; }
    } else if (which == SCR_DBG_POINT_LIST) {
        if (!binning_buf) return fail("binning_buf is NULL");
        src = bin_view((void*)binning_buf, I, 0).point_list;
        bytes = (size_t)I * 4;
    } else if (which == SCR_DBG_QMASK || which == SCR_DBG_GM_INDEX) {
        if (!binning_buf) return fail("binning_buf is NULL");
        BinView bv = bin_view((void*)binning_buf, I, 0);
        if (which == SCR_DBG_GM_INDEX && deep_lists(I, g.tiles))
            return fail("gm_index is not materialised for deep tile lists (I > 8192 per tile on average)");
        src = which == SCR_DBG_QMASK ? (const void*)bv.qmask : (const void*)bv.gm_index;
        bytes = which == SCR_DBG_QMASK ? (size_t)I : (size_t)I * 4;
    } else if (which == SCR_DBG_N_CONTRIB || which == SCR_DBG_FINAL_T) {
        if (!image_buf) return fail("image_buf is NULL");
        ImgView iv = img_view((void*)image_buf, H, W);
        src = which == SCR_DBG_N_CONTRIB ? (const void*)iv.n_contrib : (const void*)iv.final_T;
        bytes = (size_t)H * W * 4;
    } else {
        return fail("unknown debug selector %d", which);
    }
    if (bytes) HIP_TRY(hipMemcpyAsync(out, src, bytes, hipMemcpyDeviceToDevice, st));
    return 0;
}

// ---- fused expansion + compaction (expand.hip)
static inline size_t expand_nwg(int64_t n) { return (size_t)((n + 1023) / 1024); }

size_t scr_expand_scratch_bytes(int64_t n) { return align_up((expand_nwg(n) + 1) * 4) + 256; }

int scr_expand_plan(int64_t n, const float* neural_opacity, void* scratch, int64_t* num_selected_host,
                    void* stream) {
    SCR_MARK_FN;
    if (!num_selected_host) return fail("num_selected_host is NULL");
    *num_selected_host = 0;
    if (n < 0) return fail("n < 0");
    if (n == 0) return 0;
    if (!neural_opacity || !scratch) return fail("NULL argument");
    if (n >= (1ll << 31)) return fail("more than 2^31 candidates");
    hipStream_t st = (hipStream_t)stream;
    uint32_t* wg = (uint32_t*)scratch;
    unsigned long long* total = (unsigned long long*)((char*)scratch + align_up((expand_nwg(n) + 1) * 4));
    Mailbox& mb = mailbox();
    const unsigned long long seq = ++mb.seq;
    { ProfScope ps_(SCR_PROF_EXPAND, st); launch_expand_count(n, neural_opacity, wg, total, mb.dev, seq, st); }
    CHECK_LAUNCH("expand_count_kernel", 0, st);
    unsigned long long t = 0;
    if (mailbox_wait(mb, seq, false, st)) {
        t = mb.host[0];
    } else {
        HIP_TRY(hipMemcpyAsync(&t, total, 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    *num_selected_host = (int64_t)t;
    return 0;
}

// mask -> index list with the expansion's count / scan / write scheme (scratch: scr_expand_scratch_bytes(n))
int scr_mask_index_plan(int64_t n, const uint8_t* mask, void* scratch, int64_t* num_set_host, void* stream) {
    SCR_MARK_FN;
    if (!num_set_host) return fail("num_set_host is NULL");
    *num_set_host = 0;
    if (n < 0) return fail("n < 0");
    if (n == 0) return 0;
    if (!mask || !scratch) return fail("NULL argument");
    if (n >= (1ll << 31)) return fail("more than 2^31 mask entries");
    hipStream_t st = (hipStream_t)stream;
    uint32_t* wg = (uint32_t*)scratch;
    unsigned long long* total = (unsigned long long*)((char*)scratch + align_up((expand_nwg(n) + 1) * 4));
    Mailbox& mb = mailbox();
    const unsigned long long seq = ++mb.seq;
    { ProfScope ps_(SCR_PROF_EXPAND, st); launch_mask_count(n, mask, wg, total, mb.dev, seq, st); }
    CHECK_LAUNCH("mask_count_kernel", 0, st);
    unsigned long long t = 0;
    if (mailbox_wait(mb, seq, false, st)) {
        t = mb.host[0];
    } else {
        HIP_TRY(hipMemcpyAsync(&t, total, 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    *num_set_host = (int64_t)t;
    return 0;
}

int scr_mask_index_run(int64_t n, const uint8_t* mask, const void* scratch, int64_t* index, int64_t* inverse, void* stream) {
    SCR_MARK_FN;
    if (n <= 0) return n < 0 ? fail("n < 0") : 0;
    if (!mask || !scratch || (!index && !inverse)) return fail("NULL argument");
    hipStream_t st = (hipStream_t)stream;
    { ProfScope ps_(SCR_PROF_EXPAND, st); launch_mask_index(n, mask, (const uint32_t*)scratch, index, inverse, st); }
    CHECK_LAUNCH("mask_index_kernel", 0, st);
    return 0;
}

int scr_expand_run(int64_t V, int32_t k, const float* neural_opacity, const float* color,
                   const float* scale_rot, const float* offsets, int32_t offsets_ld, const float* grid_scaling,
                   const float* anchor, const void* scratch, int32_t* out_index, uint8_t* mask_out,
                   float* xyz, float* color_out, float* opacity, float* scaling, float* rot, void* stream) {
    SCR_MARK_FN;
    if (V < 0 || k <= 0) return fail("bad V / k");
    if (offsets_ld < 3 * k) return fail("offsets_ld %d: the offset rows of an anchor are 3 k = %d floats long", offsets_ld, 3 * k);
    if (V == 0) return 0;
    if (!neural_opacity || !color || !scale_rot || !offsets || !grid_scaling || !anchor || !scratch || !out_index)
        return fail("NULL argument");
    hipStream_t st = (hipStream_t)stream;
    { ProfScope ps_(SCR_PROF_EXPAND, st);
      launch_expand_run(V * k, k, neural_opacity, color, scale_rot, offsets, offsets_ld, grid_scaling, anchor,
                        (const uint32_t*)scratch, out_index, mask_out, xyz, color_out, opacity, scaling, rot, st); }
    CHECK_LAUNCH("expand_run_kernel", 0, st);
    return 0;
}

int scr_expand_backward(int64_t V, int32_t k, const float* scale_rot, const float* offsets, int32_t offsets_ld,
                        const float* grid_scaling, const int32_t* out_index, const float* g_xyz,
                        const float* g_color, const float* g_opacity, const float* g_scaling,
                        const float* g_rot, float* d_neural_opacity, float* d_color, float* d_scale_rot,
                        float* d_offsets, float* d_grid_scaling, float* d_anchor, const float* g_reg, int64_t P,
                        void* stream) {
    SCR_MARK_FN;
    if (V < 0 || k <= 0) return fail("bad V / k");
    if (g_reg && P <= 0) return fail("scr_expand_backward: g_reg needs P = the number of selected candidates");
    if (offsets_ld < 3 * k) return fail("offsets_ld %d: the offset rows of an anchor are 3 k = %d floats long", offsets_ld, 3 * k);
    if (V == 0) return 0;
    if (!scale_rot || !offsets || !grid_scaling || !out_index || !d_neural_opacity || !d_color || !d_scale_rot ||
        !d_offsets || !d_grid_scaling || !d_anchor)
        return fail("NULL argument");
    hipStream_t st = (hipStream_t)stream;
    { ProfScope ps_(SCR_PROF_EXPAND_BACKWARD, st);
      launch_expand_backward(V, k, scale_rot, offsets, offsets_ld, grid_scaling, out_index, g_xyz, g_color, g_opacity, g_scaling,
                             g_rot, d_neural_opacity, d_color, d_scale_rot, d_offsets, d_grid_scaling, d_anchor, g_reg, P, st); }
    CHECK_LAUNCH("expand_backward_kernel", 0, st);
    return 0;
}

// ---- tri-plane sampling backward (triplane.hip)
size_t scr_plane_sample_scratch_bytes(int64_t V, int32_t A, int32_t B, int32_t channels) {
    return triplane_scratch_bytes(V, A, B, channels);
}

int scr_plane_sample_backward(int64_t V, const float* coords, int32_t cstride, int32_t cx, int32_t cy, int32_t R,
                              int32_t A, int32_t B, int32_t planes, const float* grad_out0, const float* grad_out1,
                              int32_t ld, float* grad_plane0, float* grad_plane1, void* scratch, void* stream) {
    SCR_MARK_FN;
    if (V < 0 || R <= 0 || A <= 1 || B <= 1) return fail("bad sizes");
    if (planes != 1 && planes != 2) return fail("planes must be 1 or 2");
    if (cstride <= 0 || cx < 0 || cy < 0 || cx >= cstride || cy >= cstride || ld < R) return fail("bad strides");
    if (!grad_plane0 || !scratch || (V > 0 && (!coords || !grad_out0))) return fail("NULL argument");
    if (planes == 2 && (!grad_plane1 || (V > 0 && !grad_out1))) return fail("NULL argument (second plane)");
    hipStream_t st = (hipStream_t)stream;
    int rc;
    { ProfScope ps_(SCR_PROF_PLANE_BACKWARD, st);
      rc = launch_plane_sample_backward(V, coords, cstride, cx, cy, R, A, B, planes, grad_out0, grad_out1, ld, grad_plane0,
                                        grad_plane1, scratch, st); }
    if (rc == 1) return fail("R = %d channels per plane exceeds the supported 16", R);
    if (rc == 2) return fail("plane %dx%d has too many 32x32 tiles for the LDS histogram", A, B);
    CHECK_LAUNCH("plane_sample_backward", 0, st);
    return 0;
}

size_t scr_triplane_backward_scratch_bytes(int64_t V, int32_t X, int32_t Y, int32_t Z, int32_t channels) {
    return triplane_backward_scratch_bytes(V, X, Y, Z, channels);
}

int scr_triplane_backward(int64_t V, const float* coords, int32_t cstride, int32_t R, int32_t X, int32_t Y, int32_t Z,
                          int32_t planes, const float* grad_out, int32_t ld, const int32_t* cols, float* const* grad_planes,
                          void* scratch, void* stream) {
    SCR_MARK_FN;
    if (V < 0 || R <= 0 || X <= 1 || Y <= 1 || Z <= 1) return fail("bad sizes");
    if (planes != 1 && planes != 2) return fail("planes must be 1 or 2");
    if (cstride < 3 || ld < R * 3 * planes) return fail("bad strides");
    if (!cols || !grad_planes || !scratch || (V > 0 && (!coords || !grad_out))) return fail("NULL argument");
    for (int q = 0; q < 3 * planes; ++q) {
        if (!grad_planes[q]) return fail("NULL plane gradient %d", q);
        if (cols[q] < 0 || cols[q] + R > ld) return fail("column block %d outside the gradient rows", q);
    }
    hipStream_t st = (hipStream_t)stream;
    int rc;
    { ProfScope ps_(SCR_PROF_PLANE_BACKWARD, st);
      rc = launch_triplane_backward(V, coords, cstride, R, X, Y, Z, planes, grad_out, ld, cols, grad_planes, scratch, st); }
    if (rc == 1) return fail("R = %d channels per plane exceeds the supported 16", R);
    if (rc == 2) return fail("a plane of %dx%dx%d has too many 32x32 tiles for the LDS histogram", X, Y, Z);
    CHECK_LAUNCH("triplane_backward", 0, st);
    return 0;
}

size_t scr_triplane_backward_multi_scratch_bytes(int64_t V, int32_t ngrids, const int32_t* R, const int32_t* X, const int32_t* Y,
                                                 const int32_t* Z) {
    if (!R || !X || !Y || !Z || ngrids < 1 || ngrids > 3) return 0;
    return triplane_multi_scratch_bytes(V, ngrids, R, X, Y, Z);
}

int scr_triplane_backward_multi(int64_t V, const float* coords, int32_t cstride, int32_t ngrids, const int32_t* R,
                                const int32_t* X, const int32_t* Y, const int32_t* Z, const int32_t* col, const float* grad_out,
                                int32_t ld, float* const* grad_planes, void* scratch, const float* nl_coef, const float* nl_dy,
                                int32_t nl_lddy, const float* nl_x, int32_t nl_ldx, void* stream) {
    SCR_MARK_FN;
    if (V < 0 || ngrids < 1 || ngrids > 3) return fail("bad sizes");
    if (!R || !X || !Y || !Z || !col || !grad_planes || !scratch || (V > 0 && (!coords || (!grad_out && !nl_coef)))) return fail("NULL argument");
    if (nl_coef && (!nl_dy || !nl_x || nl_lddy < 32 || nl_ldx < col[ngrids - 1] + 3 * R[ngrids - 1])) return fail("fused dx: dy [V,32] and x (the sampled matrix) are needed");
    for (int g = 0; g < ngrids; ++g) {
        if (R[g] < 1 || X[g] < 2 || Y[g] < 2 || Z[g] < 2 || col[g] < 0 || col[g] + 3 * R[g] > (grad_out ? ld : nl_ldx)) return fail("bad grid %d", g);
        for (int q = 0; q < 3; ++q)
            if (!grad_planes[3 * g + q]) return fail("NULL plane gradient %d", 3 * g + q);
    }
    hipStream_t st = (hipStream_t)stream;
    int rc;
    { ProfScope ps_(SCR_PROF_PLANE_BACKWARD, st);
      rc = launch_triplane_backward_multi(V, coords, cstride, ngrids, R, X, Y, Z, col, grad_out, ld, grad_planes, scratch, nl_coef, nl_dy,
                                          nl_lddy, nl_x, nl_ldx, st); }
    if (rc == 3) return 3;      // not a layout of the fused pass: not an error, the caller goes grid by grid
    CHECK_LAUNCH("triplane_backward_multi", 0, st);
    return 0;
}

int scr_plane_row_pairs(int32_t R, int32_t A, int32_t B, const float* plane, float* pairs, void* stream) {
    SCR_MARK_FN;
    if (A < 2 || B < 2) return fail("bad sizes");
    if (!plane || !pairs) return fail("NULL argument");
    if (launch_plane_row_pairs(R, A, B, plane, pairs, (hipStream_t)stream)) return fail("R = %d channels per plane exceeds the supported 16", R);
    CHECK_LAUNCH("plane_row_pairs_kernel", 0, (hipStream_t)stream);
    return 0;
}

int scr_triplane_forward(int64_t V, const float* coords, int32_t cstride, const float* xy, const float* xz,
                         const float* yz, int32_t R, int32_t X, int32_t Y, int32_t Z, int32_t channel_last, float* out,
                         int32_t ld, int32_t col_xy, int32_t col_xz, int32_t col_yz, void* stream) {
    SCR_MARK_FN;
    if (V < 0 || R <= 0 || X <= 1 || Y <= 1 || Z <= 1) return fail("bad sizes");
    if (cstride < 3 || col_xy < 0 || col_xz < 0 || col_yz < 0 || col_xy + R > ld || col_xz + R > ld || col_yz + R > ld)
        return fail("bad strides");
    if (V > 0 && (!coords || !xy || !xz || !yz || !out)) return fail("NULL argument");
    hipStream_t st = (hipStream_t)stream;
    int rc;
    { ProfScope ps_(SCR_PROF_TRIPLANE_FORWARD, st);
      rc = launch_triplane_forward(V, coords, cstride, xy, xz, yz, R, X, Y, Z, channel_last, out, ld, col_xy, col_xz, col_yz, st); }
    if (rc == 1) return fail("R = %d channels per plane exceeds the supported 16", R);
    CHECK_LAUNCH("triplane_forward_kernel", 0, st);
    return 0;
}

// ---- plane attention of the level-0 grid (attention.hip)
size_t scr_tpa_scratch_bytes(int32_t R, int32_t H, int32_t W) { return tpa_scratch_bytes(R, H, W); }

static int tpa_bad(int32_t R, int32_t H, int32_t W) {
    if (R <= 0 || 3 * R > 24) return fail("plane attention: 3 R stacked channels must be 3..24");
    if (H <= 0 || W <= 0 || (int64_t)H * W >= (1ll << 31)) return fail("plane attention: bad plane size");
    return 0;
}

int scr_tpa_stats(int32_t R, int32_t H, int32_t W, const float* p0, const float* p1, const float* p2, float* avg,
                  float* mx, int32_t* arg, void* scratch, void* stream) {
    SCR_MARK_FN;
    if (tpa_bad(R, H, W)) return 1;
    if (!p0 || !p1 || !p2 || !avg || !mx || !arg || !scratch) return fail("NULL argument");
    hipStream_t st = (hipStream_t)stream;
    { ProfScope ps_(SCR_PROF_PLANE_ATTENTION, st); launch_tpa_stats(R, (int64_t)H * W, p0, p1, p2, avg, mx, arg, scratch, st); }
    CHECK_LAUNCH("tpa_stats_kernels", 0, st);
    return 0;
}

int scr_tpa_forward(int32_t R, int32_t H, int32_t W, const float* p0, const float* p1, const float* p2, const float* ca,
                    const float* w, float* s, uint8_t* am, float* sa, float* pair0, float* pair1, float* pair2, void* stream) {
    SCR_MARK_FN;
    if (tpa_bad(R, H, W)) return 1;
    if (!p0 || !p1 || !p2 || !ca || !w || !s || !am || !sa || !pair0 || !pair1 || !pair2) return fail("NULL argument");
    hipStream_t st = (hipStream_t)stream;
    { ProfScope ps_(SCR_PROF_PLANE_ATTENTION, st); launch_tpa_forward(R, H, W, p0, p1, p2, ca, w, s, am, sa, pair0, pair1, pair2, st); }
    CHECK_LAUNCH("tpa_apply_kernel", 0, st);
    return 0;
}

int scr_tpa_backward(int32_t R, int32_t H, int32_t W, const float* p0, const float* p1, const float* p2, const float* ca,
                     const float* w, const float* s, const uint8_t* am, const float* sa, const float* g0, const float* g1,
                     const float* g2, float* d0, float* d1, float* d2, float* dca, float* dw, void* scratch, void* stream) {
    SCR_MARK_FN;
    if (tpa_bad(R, H, W)) return 1;
    if (!p0 || !p1 || !p2 || !ca || !w || !s || !am || !sa || !g0 || !g1 || !g2 || !d0 || !d1 || !d2 || !dca || !dw || !scratch)
        return fail("NULL argument");
    hipStream_t st = (hipStream_t)stream;
    { ProfScope ps_(SCR_PROF_PLANE_ATTENTION, st);
      launch_tpa_backward(R, H, W, p0, p1, p2, ca, w, s, am, sa, g0, g1, g2, d0, d1, d2, dca, dw, scratch, st); }
    CHECK_LAUNCH("tpa_bwd_kernels", 0, st);
    return 0;
}

int scr_tpa_backward_stats(int32_t R, int32_t H, int32_t W, const float* davg, const float* dmax, const int32_t* arg,
                           float* d0, float* d1, float* d2, void* stream) {
    SCR_MARK_FN;
    if (tpa_bad(R, H, W)) return 1;
    if (!davg || !dmax || !arg || !d0 || !d1 || !d2) return fail("NULL argument");
    hipStream_t st = (hipStream_t)stream;
    { ProfScope ps_(SCR_PROF_PLANE_ATTENTION, st); launch_tpa_backward_stats(R, (int64_t)H * W, davg, dmax, arg, d0, d1, d2, st); }
    CHECK_LAUNCH("tpa_bwd_stats_kernel", 0, st);
    return 0;
}

// ---- fused L1 + SSIM loss (ssim.hip)
size_t scr_l1_ssim_scratch_bytes(int32_t C, int32_t H, int32_t W, int32_t with_grad) {
    return l1_ssim_scratch_bytes(C, H, W, with_grad);
}

int scr_l1_ssim_forward(int32_t C, int32_t H, int32_t W, const float* img1, const float* img2, void* scratch,
                        int32_t with_grad, float* out2, void* stream) {
    SCR_MARK_FN;
    if (C <= 0 || H <= 0 || W <= 0) return fail("bad image size");
    if (!img1 || !img2 || !scratch || !out2) return fail("NULL argument");
    hipStream_t st = (hipStream_t)stream;
    { ProfScope ps_(SCR_PROF_L1_SSIM, st); launch_l1_ssim_forward(C, H, W, img1, img2, scratch, with_grad, out2, st); }
    CHECK_LAUNCH("l1_ssim_forward_kernel", 0, st);
    return 0;
}

int scr_l1_ssim_backward(int32_t C, int32_t H, int32_t W, const float* img1, const float* img2,
                         const void* scratch, const float* g_l1, const float* g_ssim, float* dimg1,
                         void* stream) {
    SCR_MARK_FN;
    if (C <= 0 || H <= 0 || W <= 0) return fail("bad image size");
    if (!img1 || !img2 || !scratch || !g_l1 || !g_ssim || !dimg1) return fail("NULL argument");
    hipStream_t st = (hipStream_t)stream;
    { ProfScope ps_(SCR_PROF_L1_SSIM_BACKWARD, st);
      launch_l1_ssim_backward(C, H, W, img1, img2, scratch, g_l1, g_ssim, dimg1, st); }
    CHECK_LAUNCH("l1_ssim_backward_kernel", 0, st);
    return 0;
}

// ---- scaling regulariser of the per-view loss (ssim.hip)
size_t scr_scaling_reg_scratch_bytes(int64_t P) { return scaling_reg_scratch_bytes(P); }

int scr_scaling_reg_forward(int64_t P, const float* scaling, void* scratch, float* out, void* stream) {
    SCR_MARK_FN;
    if (P <= 0) return fail("P <= 0");
    if (!scaling || !scratch || !out) return fail("NULL argument");
    hipStream_t st = (hipStream_t)stream;
    { ProfScope ps_(SCR_PROF_L1_SSIM, st); launch_scaling_reg_forward(P, scaling, scratch, out, st); }
    CHECK_LAUNCH("scaling_reg_partial_kernel", 0, st);
    return 0;
}

int scr_scaling_reg_backward(int64_t P, const float* scaling, const float* g, float* dscaling, void* stream) {
    SCR_MARK_FN;
    if (P <= 0) return fail("P <= 0");
    if (!scaling || !g || !dscaling) return fail("NULL argument");
    hipStream_t st = (hipStream_t)stream;
    { ProfScope ps_(SCR_PROF_L1_SSIM_BACKWARD, st); launch_scaling_reg_backward(P, scaling, g, dscaling, st); }
    CHECK_LAUNCH("scaling_reg_backward_kernel", 0, st);
    return 0;
}

// ---- the L1 of the cross-view consistency term (ssim.hip)
size_t scr_pair_l1_scratch_bytes(int64_t n) { return pair_l1_scratch_bytes(n > 0 ? n : 1); }

int scr_pair_l1_forward(int64_t n, const float* gen1, const float* gen2, const float* real1, const float* real2, void* scratch,
                        float* out, void* stream) {
    SCR_MARK_FN;
    if (n <= 0) return fail("n <= 0");
    if (!gen1 || !gen2 || !real1 || !real2 || !scratch || !out) return fail("NULL argument");
    hipStream_t st = (hipStream_t)stream;
    { ProfScope ps_(SCR_PROF_L1_SSIM, st); launch_pair_l1_forward(n, gen1, gen2, real1, real2, scratch, out, st); }
    CHECK_LAUNCH("pair_l1_partial_kernel", 0, st);
    return 0;
}

int scr_pair_l1_backward(int64_t n, const float* gen1, const float* gen2, const float* real1, const float* real2, const float* g,
                         float* d_gen1, float* d_gen2, void* stream) {
    SCR_MARK_FN;
    if (n <= 0) return fail("n <= 0");
    if (!gen1 || !gen2 || !real1 || !real2 || !g) return fail("NULL argument");
    if (!d_gen1 && !d_gen2) return 0;
    hipStream_t st = (hipStream_t)stream;
    { ProfScope ps_(SCR_PROF_L1_SSIM_BACKWARD, st); launch_pair_l1_backward(n, gen1, gen2, real1, real2, g, d_gen1, d_gen2, st); }
    CHECK_LAUNCH("pair_l1_backward_kernel", 0, st);
    return 0;
}

// ---- visible-anchor gather (anchor_gather.hip)
int32_t scr_anchor_gather_stat_rows(int64_t V) { return anchor_gather_stat_rows(V > 0 ? V : 1); }
int64_t scr_anchor_gather_stat_buffer_rows(int64_t V) { return anchor_gather_stat_buffer_rows(V > 0 ? V : 1); }

int scr_anchor_gather(int64_t V, const int64_t* visible_index, const float* anchor_feat, const float* anchor,
                      const float* offset, const float* scaling, float* feat_out, float* anchor_out, float* offsets_out,
                      float* grid_scaling_out, float* g_fea_out, int32_t g_fea_ld, float* col_stats_out, void* stream) {
    SCR_MARK_FN;
    if (V < 0) return fail("V < 0");
    if (g_fea_ld != 71 && g_fea_ld != 72) return fail("g_fea row stride must be 71 (packed) or 72 (16-byte aligned rows)");
    if (V == 0) return 0;
    if (!visible_index || !anchor_feat || !anchor || !offset || !scaling || !anchor_out || !grid_scaling_out || !g_fea_out)
        return fail("NULL argument");
    if ((!feat_out || !offsets_out) && g_fea_ld != 72)
        return fail("feat_out / offsets_out may be NULL (their consumers read the columns of g_fea) only with 16-byte aligned g_fea rows (ld 72)");
    launch_anchor_gather(V, visible_index, anchor_feat, anchor, offset, scaling, feat_out, anchor_out, offsets_out,
                         grid_scaling_out, g_fea_out, g_fea_ld, col_stats_out, (hipStream_t)stream);
    CHECK_LAUNCH("anchor_gather_kernel", 0, (hipStream_t)stream);
    return 0;
}

int scr_anchor_gather_backward(int64_t N, int64_t V, const int64_t* inverse_index, const float* grid_scaling, const float* d_feat,
                               const float* d_anchor, const float* d_offsets, const float* d_grid_scaling,
                               const float* d_g_fea, int32_t g_fea_ld, float* g_anchor_feat, float* g_anchor,
                               float* g_offset, float* g_scaling, int32_t accumulate, const float* nl_coef, const float* nl_dy,
                               int32_t nl_lddy, const float* nl_x, int32_t nl_ldx, void* stream) {
    SCR_MARK_FN;
    if (N < 0 || V < 0) return fail("bad sizes");      // V may exceed N: N can be a range of the anchors (see the header)
    if (g_fea_ld != 71 && g_fea_ld != 72) return fail("g_fea row stride must be 71 (packed) or 72 (16-byte aligned rows)");
    if (N == 0) return 0;
    if (!inverse_index || !g_anchor_feat || !g_anchor || !g_offset || !g_scaling) return fail("NULL argument");
    if (d_grid_scaling || d_g_fea || nl_coef) { if (!grid_scaling) return fail("grid_scaling is needed for d exp"); }
    if (nl_coef) {
        if (!nl_dy || !nl_x || nl_lddy < 32 || nl_ldx < 71) return fail("fused dx: dy [V,32] and x = g_fea [V,71] are needed");
        if (nl_lddy % 4 != 0 || ((uintptr_t)nl_dy & 15) != 0 || ((uintptr_t)nl_coef & 15) != 0)
            return fail("fused dx: dy and the coefficients must be 16-byte aligned, dy's row stride a multiple of 4 floats");
    }
    launch_anchor_gather_backward(N, V, inverse_index, grid_scaling, d_feat, d_anchor, d_offsets, d_grid_scaling, d_g_fea, g_fea_ld,
                                  g_anchor_feat, g_anchor, g_offset, g_scaling, accumulate, nl_coef, nl_dy, nl_lddy, nl_x, nl_ldx,
                                  (hipStream_t)stream);
    CHECK_LAUNCH("anchor_gather_backward_kernel", 0, (hipStream_t)stream);
    return 0;
}

// ---- normalised tri-plane coordinates of contiguous xyz[V,3] in the box [lo, hi] (triplane.hip)
int scr_box_coords(int64_t V, const float* xyz, const float* lo_host, const float* hi_host, float* out, void* stream) {
    SCR_MARK_FN;
    if (V < 0) return fail("V < 0");
    if (V == 0) return 0;
    if (!xyz || !lo_host || !hi_host || !out) return fail("NULL argument");
    launch_box_coords(V, xyz, lo_host, hi_host, out, (hipStream_t)stream);
    CHECK_LAUNCH("box_coords_kernel", 0, (hipStream_t)stream);
    return 0;
}

// ---- the parameter side of the fold: G, c from the pairs' weights, their gradients back, running statistics (normlinear.hip)
int scr_norm_fold(int32_t L, int32_t d, const int32_t* widths_host, const int32_t* cols_host, const uint8_t* col_at_host,
                  const void* const* lin_weight_host,
                  const void* const* lin_bias_host, const void* const* bn_weight_host, const void* const* bn_bias_host, float* G,
                  float* c, void* stream) {
    SCR_MARK_FN;
    if (!widths_host || !cols_host || !lin_weight_host || !lin_bias_host || !bn_weight_host || !bn_bias_host || !G || !c)
        return fail("NULL argument");
    if (launch_nl_fold(L, d, widths_host, cols_host, col_at_host, (const float* const*)lin_weight_host, (const float* const*)lin_bias_host,
                       (const float* const*)bn_weight_host, (const float* const*)bn_bias_host, G, c, (hipStream_t)stream))
        return fail("scr_norm_fold: 1 <= L <= 4 pairs, column blocks inside [0, d), d <= 80, col_at a permutation of 0 .. d-1");
    CHECK_LAUNCH("nl_fold_kernel", 0, (hipStream_t)stream);
    return 0;
}

int scr_norm_fold_backward(int32_t L, int32_t d, const int32_t* widths_host, const int32_t* cols_host, const uint8_t* col_at_host,
                           const void* const* lin_weight_host, const void* const* bn_weight_host, const void* const* bn_bias_host,
                           const float* dG, const float* dc, void* const* d_lin_weight_host, void* const* d_lin_bias_host,
                           void* const* d_bn_weight_host, void* const* d_bn_bias_host, void* stream) {
    SCR_MARK_FN;
    if (!widths_host || !cols_host || !lin_weight_host || !bn_weight_host || !bn_bias_host || !dG || !dc || !d_lin_weight_host ||
        !d_lin_bias_host || !d_bn_weight_host || !d_bn_bias_host)
        return fail("NULL argument");
    if (launch_nl_fold_backward(L, d, widths_host, cols_host, col_at_host, (const float* const*)lin_weight_host,
                                (const float* const*)bn_weight_host, (const float* const*)bn_bias_host, dG, dc,
                                (float* const*)d_lin_weight_host, (float* const*)d_lin_bias_host, (float* const*)d_bn_weight_host,
                                (float* const*)d_bn_bias_host, (hipStream_t)stream))
        return fail("scr_norm_fold_backward: 1 <= L <= 4 pairs, column blocks inside [0, d), d <= 80");
    CHECK_LAUNCH("nl_fold_backward_kernel", 0, (hipStream_t)stream);
    return 0;
}

int scr_norm_running_stats(int32_t L, int32_t d, const int32_t* widths_host, const int32_t* cols_host, const uint8_t* col_at_host,
                           const float* momentum_host,
                           void* const* running_mean_host, void* const* running_var_host, void* const* num_batches_host,
                           const float* mean, const float* var, int64_t n, void* stream) {
    SCR_MARK_FN;
    if (!widths_host || !cols_host || !momentum_host || !running_mean_host || !running_var_host || !mean || !var)
        return fail("NULL argument");
    if (launch_nl_running_stats(L, d, widths_host, cols_host, col_at_host, momentum_host, (float* const*)running_mean_host,
                                (float* const*)running_var_host, (long long* const*)num_batches_host, mean, var, n,
                                (hipStream_t)stream))
        return fail("scr_norm_running_stats: 1 <= L <= 4");
    CHECK_LAUNCH("nl_running_stats_kernel", 0, (hipStream_t)stream);
    return 0;
}

// ---- BatchNorm1d (batch statistics) folded into Linear(d, 32) (normlinear.hip)
size_t scr_norm_linear_scratch_bytes(int64_t V) { return norm_linear_scratch_bytes(V > 0 ? V : 1); }

int scr_norm_linear_forward(int64_t V, int32_t d, const float* x, int32_t ldx, const float* G, const float* c, float eps,
                            float* y, float* mean, float* var, float* inv, void* scratch, const float* col_stats,
                            int32_t col_stat_rows, void* stream) {
    SCR_MARK_FN;
    if (V < 1 || d < 1 || ldx < d) return fail("bad sizes");
    if ((col_stats != nullptr) != (col_stat_rows > 0)) return fail("col_stats and col_stat_rows go together (NULL / 0: the statistics pass runs here)");
    if (!x || !G || !c || !y || !mean || !var || !inv || !scratch) return fail("NULL argument");
    hipStream_t st = (hipStream_t)stream;
    int rc;
    { ProfScope ps_(SCR_PROF_NORM_LINEAR, st);
      rc = launch_norm_linear_forward(V, d, x, ldx, G, c, eps, y, mean, var, inv, scratch, col_stats, col_stat_rows, st); }
    if (rc == 1) return fail("d = %d input columns exceed the supported 80", d);
    CHECK_LAUNCH("norm_linear_forward", 0, st);
    return 0;
}

int scr_norm_linear_backward(int64_t V, int32_t d, const float* x, int32_t ldx, const float* dy, int32_t lddy, const float* G,
                             const float* mean, const float* inv, float* dx, int32_t lddx, float* dG, float* dc,
                             void* scratch, float* coef_out, void* stream) {
    SCR_MARK_FN;
    if (V < 1 || d < 1 || ldx < d || lddy < 32 || (dx && lddx < d)) return fail("bad sizes");
    if (!x || !dy || !G || !mean || !inv || !dG || !dc || !scratch) return fail("NULL argument");
    hipStream_t st = (hipStream_t)stream;
    int rc;
    { ProfScope ps_(SCR_PROF_NORM_LINEAR_BACKWARD, st);
      rc = launch_norm_linear_backward(V, d, x, ldx, dy, lddy, G, mean, inv, dx, lddx, dG, dc, scratch, coef_out, st); }
    if (rc == 3) return fail("copy of the backward coefficients failed: %s", hipGetErrorString(hipGetLastError()));
    if (rc == 1) return fail("d = %d input columns exceed the supported 80", d);
    if (rc == 2) return fail("dy must be 16-byte aligned with a row stride that is a multiple of 4 floats");
    CHECK_LAUNCH("norm_linear_backward", 0, st);
    return 0;
}

int scr_norm_linear_dx(int64_t V, int32_t d, const float* x, int32_t ldx, const float* dy, int32_t lddy, const float* coef,
                       float* dx, int32_t lddx, void* stream) {
    SCR_MARK_FN;
    if (V < 1 || d < 1 || ldx < d || lddy < 32 || lddx < d) return fail("bad sizes");
    if (!x || !dy || !coef || !dx) return fail("NULL argument");
    hipStream_t st = (hipStream_t)stream;
    int rc;
    { ProfScope ps_(SCR_PROF_NORM_LINEAR_BACKWARD, st);
      rc = launch_norm_linear_dx(V, d, x, ldx, dy, lddy, coef, dx, lddx, st); }
    if (rc == 1) return fail("d = %d input columns exceed the supported 80", d);
    if (rc == 2) return fail("dy must be 16-byte aligned with a row stride that is a multiple of 4 floats");
    CHECK_LAUNCH("norm_linear_dx", 0, st);
    return 0;
}

// ---- MLP heads (mlp_heads.hip)
size_t scr_mlp_heads_hidden_bytes(int64_t V) { return mlp_heads_hidden_bytes(V > 0 ? V : 1); }
size_t scr_mlp_heads_partial_bytes(int64_t V) { return mlp_heads_partial_bytes(V > 0 ? V : 1); }

int scr_mlp_heads_forward(int64_t V, const float* feat, int32_t feat_ld, const float* anchor, const float* campos, const float* geo_a, const float* geo_b,
                          const float* w1, const float* b1, const float* w2o, const float* b2o, const float* w2c,
                          const float* b2c, const float* w2v, const float* b2v, void* hidden_save, float* out_opacity,
                          float* out_color, float* out_cov, void* stream) {
    SCR_MARK_FN;
    if (V < 0) return fail("V < 0");
    if (feat_ld < 32 || feat_ld % 4 != 0 || ((uintptr_t)feat & 15) != 0) return fail("feat rows: 32 floats, 16-byte aligned, feat_ld a multiple of 4 (got %d)", feat_ld);
    if (V == 0) return 0;
    if (!feat || !anchor || !campos || !geo_a || !geo_b || !w1 || !b1 || !w2o || !b2o || !w2c || !b2c || !w2v || !b2v || !hidden_save ||
        !out_opacity || !out_color || !out_cov)
        return fail("NULL argument");
    hipStream_t st = (hipStream_t)stream;
    { ProfScope ps_(SCR_PROF_MLP_HEADS, st);
      launch_mlp_heads_forward(V, feat, feat_ld, anchor, campos, geo_a, geo_b, w1, b1, w2o, b2o, w2c, b2c, w2v, b2v, hidden_save, out_opacity,
                               out_color, out_cov, st); }
    CHECK_LAUNCH("mlp_heads_forward_kernel", 0, st);
    return 0;
}

int scr_mlp_heads_backward(int64_t V, const float* feat, int32_t feat_ld, const float* anchor, const float* campos, const float* geo_a, const float* geo_b,
                           const float* w1, const float* w2o, const float* w2c, const float* w2v, const void* hidden_save,
                           const float* out_opacity, const float* out_color, const float* g_opacity, const float* g_color,
                           const float* g_cov, void* partial, float* d_feat, float* d_anchor, float* d_geo_a, float* d_geo_b, float* d_w1,
                           float* d_b1, float* d_w2o, float* d_b2o, float* d_w2c, float* d_b2c, float* d_w2v, float* d_b2v,
                           void* stream) {
    SCR_MARK_FN;
    if (V <= 0) return fail("V <= 0");
    if (feat_ld < 32 || feat_ld % 4 != 0 || ((uintptr_t)feat & 15) != 0) return fail("feat rows: 32 floats, 16-byte aligned, feat_ld a multiple of 4 (got %d)", feat_ld);
    if (!feat || !anchor || !campos || !geo_a || !geo_b || !w1 || !w2o || !w2c || !w2v || !hidden_save || !out_opacity || !out_color ||
        !g_opacity || !g_color || !g_cov || !partial || !d_feat || !d_anchor || !d_geo_a || !d_geo_b || !d_w1 || !d_b1 || !d_w2o || !d_b2o ||
        !d_w2c || !d_b2c || !d_w2v || !d_b2v)
        return fail("NULL argument");
    hipStream_t st = (hipStream_t)stream;
    { ProfScope ps_(SCR_PROF_MLP_HEADS_BACKWARD, st);
      launch_mlp_heads_backward(V, feat, feat_ld, anchor, campos, geo_a, geo_b, w1, w2o, w2c, w2v, hidden_save, out_opacity, out_color, g_opacity,
                                g_color, g_cov, partial, d_feat, d_anchor, d_geo_a, d_geo_b, d_w1, d_b1, d_w2o, d_b2o, d_w2c, d_b2c, d_w2v,
                                d_b2v, st); }
    CHECK_LAUNCH("mlp_heads_backward_kernel", 0, st);
    return 0;
}

// ---- densification statistics (densify.hip)
int scr_statis_compute(int64_t V, int32_t k, const float* neural_opacity, const int32_t* out_index,
                       const uint8_t* update_filter, const float* viewspace_grad, int32_t grad_stride,
                       float* inc_opacity, float* inc_grad, void* stream) {
    SCR_MARK_FN;
    if (V < 0 || k <= 0 || k > 256 || grad_stride < 2) return fail("bad V / k (1..256) / grad_stride");
    if (V == 0) return 0;
    if (!neural_opacity || !out_index || !inc_opacity || !inc_grad) return fail("NULL argument");
    hipStream_t st = (hipStream_t)stream;
    launch_statis_compute(V, k, neural_opacity, out_index, update_filter, viewspace_grad, grad_stride, inc_opacity,
                          inc_grad, st);
    CHECK_LAUNCH("statis_compute_kernel", 0, st);
    return 0;
}

int scr_statis_apply(int64_t V, int32_t k, const int64_t* visible_index, const float* inc_opacity, const float* inc_grad,
                     float* opacity_accum, float* anchor_demon, float* offset_gradient_accum, float* offset_denom,
                     void* stream) {
    SCR_MARK_FN;
    if (V < 0 || k <= 0) return fail("bad V / k");
    if (V == 0) return 0;
    if (!visible_index || !inc_opacity || !inc_grad || !opacity_accum || !anchor_demon || !offset_gradient_accum ||
        !offset_denom)
        return fail("NULL argument");
    hipStream_t st = (hipStream_t)stream;
    launch_statis_apply(V, k, visible_index, inc_opacity, inc_grad, opacity_accum, anchor_demon, offset_gradient_accum,
                        offset_denom, st);
    CHECK_LAUNCH("statis_apply_kernel", 0, st);
    return 0;
}

// ---- optimizer step (adam.hip)
int scr_adam_step(int32_t n_tensors, const scr_adam_tensor* tensors, double beta1, double beta2, double eps, void* stream) {
    SCR_MARK_FN;
    if (n_tensors < 0) return fail("scr_adam_step: n_tensors < 0");
    if (n_tensors == 0) return 0;
    if (!tensors) return fail("scr_adam_step: tensors is NULL");
    if (!(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.0)) return fail("scr_adam_step: bad beta / eps");
    for (int t = 0; t < n_tensors; ++t) {
        const scr_adam_tensor& x = tensors[t];
        if (x.numel < 0 || (x.numel > 0 && (!x.param || !x.grad || !x.exp_avg || !x.exp_avg_sq))) return fail("scr_adam_step: NULL tensor");
        if (!(x.step_size >= 0.0) || !(x.step_size < 1e30) || !(x.bias_correction2_sqrt > 0.0))
            return fail("scr_adam_step: step_size must be finite and >= 0, bias_correction2_sqrt > 0 (step >= 1)");
    }
    hipStream_t st = (hipStream_t)stream;
    if (launch_adam(n_tensors, tensors, beta1, beta2, eps, st)) return fail("scr_adam_step: more than 2^31 workgroups in one launch");
    CHECK_LAUNCH("adam_kernel", 0, st);
    return 0;
}

// ---- tri-plane total-variation term (tv.hip)
int scr_tv_add_grad(int32_t n_planes, const scr_tv_plane* planes, void* stream) {
    SCR_MARK_FN;
    if (n_planes < 0) return fail("scr_tv_add_grad: n_planes < 0");
    if (n_planes == 0) return 0;
    if (!planes) return fail("scr_tv_add_grad: planes is NULL");
    for (int t = 0; t < n_planes; ++t) {
        const scr_tv_plane& x = planes[t];
        if (x.channels < 0 || x.rows < 0 || x.cols < 0) return fail("scr_tv_add_grad: negative plane size");
        if ((int64_t)x.channels * x.rows * x.cols > 0 && (!x.plane || !x.grad)) return fail("scr_tv_add_grad: NULL plane / grad");
        if (x.plane == x.grad && x.plane) return fail("scr_tv_add_grad: plane and grad must be distinct buffers");
    }
    hipStream_t st = (hipStream_t)stream;
    if (launch_tv_add_grad(n_planes, planes, st)) return fail("scr_tv_add_grad: more than 2^31 wave units in one launch");
    CHECK_LAUNCH("tv_add_grad_kernel", 0, st);
    return 0;
}

int scr_knn(int64_t N, int32_t k, const float* grid_host, const float* sorted_pts, const int64_t* sorted_id,
            const int32_t* cell_start, int64_t* out_idx, void* stream) {
    SCR_MARK_FN;
    if (N <= 0 || k <= 0 || k > 16) return fail("scr_knn: need N > 0 and 1 <= k <= 16");
    if (N <= k) return fail("scr_knn: fewer than k + 1 points");
    if (!grid_host || !sorted_pts || !sorted_id || !cell_start || !out_idx) return fail("NULL argument");
    if (!(grid_host[3] > 0.0f) || grid_host[4] < 1 || grid_host[5] < 1 || grid_host[6] < 1) return fail("bad grid");
    launch_knn(N, k, grid_host, sorted_pts, sorted_id, cell_start, out_idx, (hipStream_t)stream);
    CHECK_LAUNCH("knn_kernel", 0, (hipStream_t)stream);
    return 0;
}

int scr_knn_curvature(int64_t N, int32_t k, const float* points, const int64_t* idx, float* curvature, void* stream) {
    SCR_MARK_FN;
    if (N <= 0 || k < 2) return fail("bad N / k");
    if (!points || !idx || !curvature) return fail("NULL argument");
    launch_knn_curvature(N, k, points, idx, curvature, (hipStream_t)stream);
    CHECK_LAUNCH("knn_curvature_kernel", 0, (hipStream_t)stream);
    return 0;
}

int scr_copy_probe(const void* src, void* dst, size_t bytes, void* stream) {
    SCR_MARK_FN;
    if (!src || !dst || bytes < 16) return fail("NULL argument");
    const size_t n = bytes / 16;
    copy_probe_kernel<<<(unsigned)((n + 1023) / 1024), 256, 0, (hipStream_t)stream>>>((const copy_f4*)src, (copy_f4*)dst, n);
    CHECK_LAUNCH("copy_probe_kernel", 0, (hipStream_t)stream);
    return 0;
}

int scr_profile_enable(int mask) {
    g_prof_mask = mask < 0 ? 0xffffffffu : (unsigned)mask;
    memset(g_prof_seen, 0, sizeof(g_prof_seen));
    return 0;
}

int scr_profile_stride(int every) {
    if (every < 1) return fail("scr_profile_stride: every >= 1");
    g_prof_stride = (unsigned)every;
    memset(g_prof_seen, 0, sizeof(g_prof_seen));
    return 0;
}

int scr_profile_read(double* total_ms, int64_t* launches) {
    if (!total_ms || !launches) return fail("NULL argument");
    for (auto& r : g_prof_recs) {
        HIP_TRY(hipEventSynchronize(r.b));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, r.a, r.b));
        total_ms[r.idx] += ms;
        launches[r.idx] += 1;
        g_prof_pool.push_back(r.a);
        g_prof_pool.push_back(r.b);
    }
    g_prof_recs.clear();
    return 0;
}

const char* scr_profile_kernel_name(int idx) { return idx >= 0 && idx < SCR_PROF_COUNT ? kProfNames[idx] : ""; }

int scr_markers_enable(int on) {
    if (!on) { g_mark_on = false; return 0; }
    if (!g_roctx_push) {
        void* h = nullptr;
        for (const char* name : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
            h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (h && dlsym(h, "roctxRangePushA") && dlsym(h, "roctxRangePop")) break;
            h = nullptr;
        }
        if (!h) return fail("scr_markers_enable: no roctx library (librocprofiler-sdk-roctx.so / libroctx64.so) can be loaded");
        g_roctx_push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
        g_roctx_pop = (int (*)())dlsym(h, "roctxRangePop");
    }
    g_mark_on = true;
    return 0;
}
int scr_marker_push(const char* name) {
    if (g_mark_on && name) (void)g_roctx_push(name);
    return 0;
}
int scr_marker_pop(void) {
    if (g_mark_on) (void)g_roctx_pop();
    return 0;
}

}  // extern "C"
