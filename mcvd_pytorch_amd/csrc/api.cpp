// extern "C" surface of libmcvd_hip.so (see include/mcvd_hip.h).  Nothing here throws.
#include <dlfcn.h>
#include <errno.h>
#include <fcntl.h>
#include <pthread.h>
#include <math.h>
#include <sys/file.h>
#include <sys/stat.h>
#include <unistd.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <mutex>
#include <new>
#include <vector>

#include "model.h"

using namespace mcvd;

#define API_TRY try {
#define API_CATCH                                             \
    }                                                         \
    catch (const std::bad_alloc&) {                           \
        mcvd::set_error("out of host memory");                \
        return MCVD_ENOMEM;                                   \
    }                                                         \
    catch (...) {                                             \
        mcvd::set_error("unexpected C++ exception");          \
        return MCVD_EINVAL;                                   \
    }

// ---- one process per GPU, enforced.  Kernels of two PROCESSES that share the CUs of one MI355X corrupted each other's results
// (profiles/r04_two_process_corruption.txt: with attn_h2_kernel<3,3> of one process resident, 20-35 % of another process's elementwise
// launches came back with 16 lanes of one VALU result wrong.  Cause, found in round 6 (profiles/r06_coresident_cause.txt): a v_pk_fma_f32 that
// reads one VGPR pair as src1 and src2 loses its low addend beside another wave's 128-bit-operand MFMA; this library's kernels no longer hold
// that form (common.h fma_unpacked, tools/check_vop3p_dual_read.py), a co-tenant's kernels may).  The design never co-schedules
// (DESIGN section 7), and this guard makes the unsupported configuration loud instead of silently wrong: every process that creates a
// context on a device holds an advisory lock  <lock dir>/mcvd_hip_gpu_<pci bus id>.lock  (flock, released by the kernel when the process
// exits, however it exits) for as long as it has a context there.  A second process is REFUSED (MCVD_EBUSY) unless MCVD_ALLOW_SHARED_DEVICE=1
// is set in ITS environment at mcvd_ctx_create (tests that let ranks take turns on one device, profiler passes spawned by an idle parent);
// with the override the context is marked shared and keeps the one kernel identified as the aggressor (the three-piece bf16 attention)
// off the device -- attention then runs on the fp32 MFMA.  Best effort by construction: processes that do not see the same lock
// directory (other containers, other users with a private /dev/shm) are not seen.
namespace {
std::mutex g_dev_mu;
struct DevLock { int fd = -1; int refs = 0; bool exclusive = false; };
DevLock g_dev_lock[64];
std::vector<mcvd_ctx*> g_live_ctx[64];       // live contexts of THIS process per device (two streams of one process: see mcvd_ctx_shares_device)

// returns 0 (this process holds the device alone), 1 (another process holds it; *we hold a shared lock*), < 0 error
int device_lock_acquire(int device) {
    std::lock_guard<std::mutex> g(g_dev_mu);
    if (device < 0 || device >= 64) return 0;
    DevLock& L = g_dev_lock[device];
    if (L.refs > 0) { ++L.refs; return L.exclusive ? 0 : 1; }
    char bus[64] = "";
    if (hipDeviceGetPCIBusId(bus, sizeof(bus), device) != hipSuccess) snprintf(bus, sizeof(bus), "dev%d", device);
    for (char* c = bus; *c; ++c)
        if (!((*c >= '0' && *c <= '9') || (*c >= 'a' && *c <= 'z') || (*c >= 'A' && *c <= 'Z'))) *c = '_';
    // The lock file has a predictable name in a world-writable directory: never follow a symlink planted there (O_NOFOLLOW), accept a
    // regular file only (fstat), and open it read-only when it belongs to another user (flock needs no write access) so that the
    // one-process-per-GPU rule holds across users too.  The guard stays ADVISORY: a local user can hold the lock and deny the device
    // (MCVD_EBUSY is loud, never wrong results), and a sharer that could not take LOCK_SH while the first process holds LOCK_EX holds
    // nothing once that process exits.
    const char* dirs[2] = {"/dev/shm", "/tmp"};
    int fd = -1;
    for (int i = 0; i < 2 && fd < 0; ++i) {
        char path[256];
        snprintf(path, sizeof(path), "%s/mcvd_hip_gpu_%s.lock", dirs[i], bus);
        fd = open(path, O_RDWR | O_CREAT | O_CLOEXEC | O_NOFOLLOW, 0666);
        if (fd < 0 && (errno == EACCES || errno == EPERM)) fd = open(path, O_RDONLY | O_CLOEXEC | O_NOFOLLOW);
        if (fd >= 0) {
            struct stat sb;
            if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode)) { close(fd); fd = -1; }
        }
    }
    if (fd < 0) return 0;                                   // no usable lock file: nothing to enforce with
    L.fd = fd;
    L.refs = 1;
    if (flock(fd, LOCK_EX | LOCK_NB) == 0) { L.exclusive = true; return 0; }
    L.exclusive = false;
    (void)flock(fd, LOCK_SH | LOCK_NB);                     // a sharer: later arrivals still see the device as taken (best effort, see above)
    return 1;
}

void device_lock_release(int device) {
    std::lock_guard<std::mutex> g(g_dev_mu);
    if (device < 0 || device >= 64) return;
    DevLock& L = g_dev_lock[device];
    if (L.refs > 0 && --L.refs == 0) {
        if (L.fd >= 0) close(L.fd);                         // drops the flock
        L = DevLock{};
    }
}
// a fork()ed child inherits the table but not the parent's claim on the device: it starts with no lock and takes its own on its first context
void device_lock_atfork_child() {
    for (DevLock& L : g_dev_lock) {
        if (L.fd >= 0) close(L.fd);                         // (closing the child's copy of the descriptor does not release the parent's flock)
        L = DevLock{};
    }
    for (auto& v : g_live_ctx) v.clear();
}
struct AtForkOnce { AtForkOnce() { pthread_atfork(nullptr, nullptr, device_lock_atfork_child); } } g_atfork_once;

void ctx_register(mcvd_ctx* c) {
    std::lock_guard<std::mutex> g(g_dev_mu);
    if (c->device < 0 || c->device >= 64) return;
    for (mcvd_ctx* o : g_live_ctx[c->device]) ++o->epoch;      // their captured graphs embed an attention kernel chosen for a device of their own
    g_live_ctx[c->device].push_back(c);
}
void ctx_unregister(mcvd_ctx* c) {
    std::lock_guard<std::mutex> g(g_dev_mu);
    if (c->device < 0 || c->device >= 64) return;
    auto& v = g_live_ctx[c->device];
    v.erase(std::remove(v.begin(), v.end(), c), v.end());
}
}  // namespace

// Does anything else run kernels on this context's device CONCURRENTLY with it?  Another process (the context was let in by
// MCVD_ALLOW_SHARED_DEVICE=1), or another live context of this process bound to a DIFFERENT stream.  Round 5 measured the second case
// (tools/diag_concurrent_streams.py, profiles/r05_two_stream_corruption.txt): two streams of ONE process corrupt each other exactly like two
// processes do -- 55 % of the elementwise launches beside attn_h2_kernel<3,3> on the other stream came back with 16 lanes of one VALU
// result wrong, none beside the fp32 attention kernel, none with the two sides on disjoint CU halves (HSA_CU_MASK).  Contexts that share
// a stream are serialised by it and do not count.  Round 6 found the cause (one instruction form in the VICTIMS, profiles/r06_coresident_cause.txt)
// and removed it from every kernel of the library, so sharing is REPORTED (mcvd_ctx_device_shared) but no longer changes which kernels run; the
// round-5 workaround -- the split-operand attention kernels kept off a shared device -- is the option "share_fence" (default 0).
bool mcvd_ctx_shares_device(const mcvd_ctx* c) {
    if (!c) return false;
    if (c->shared_device) return true;
    // (The context's own second stream, option "side_stream", overlaps only kernels of THIS library, none of which holds the instruction form that
    // breaks beside a 128-bit-operand MFMA -- checked on the linked library by every build, tools/check_vop3p_dual_read.py: it does not count.)
    std::lock_guard<std::mutex> g(g_dev_mu);
    if (c->device < 0 || c->device >= 64) return false;
    for (const mcvd_ctx* o : g_live_ctx[c->device])
        if (o != c && o->stream != c->stream) return true;
    return false;
}

extern "C" {

const char* mcvd_version(void) { return "mcvd_hip 0.1 (gfx950)"; }

const char* mcvd_last_error(mcvd_ctx*) { return mcvd::get_error(); }

int mcvd_ctx_create(int device, void* hip_stream, mcvd_ctx** out) {
    API_TRY
    MCVD_REQUIRE(out, "ctx_create: out is NULL");
    int n = 0;
    MCVD_HIP_CHECK(hipGetDeviceCount(&n));
    MCVD_REQUIRE(device >= 0 && device < n, "ctx_create: device %d not in [0,%d)", device, n);
    MCVD_HIP_CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    MCVD_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    MCVD_REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0, "this library is built for gfx950 only; device %d is %s", device,
                 prop.gcnArchName);
    const int taken = device_lock_acquire(device);
    bool allow_shared = false;
    if (const char* t = getenv("MCVD_ALLOW_SHARED_DEVICE")) allow_shared = atoi(t) != 0;
    if (taken == 1 && !allow_shared) {
        device_lock_release(device);
        set_error("ctx_create: device %d is in use by ANOTHER PROCESS of this library.  One process per GPU is the supported configuration: "
                  "co-resident kernels of two processes corrupted each other's results on MI355X (profiles/r04_two_process_corruption.txt).  "
                  "Give each process its own GPU, or set MCVD_ALLOW_SHARED_DEVICE=1 if the processes take turns on the device", device);
        return MCVD_EBUSY;
    }
    mcvd_ctx* c = new mcvd_ctx();
    c->device = device;
    c->shared_device = taken == 1 ? 1 : 0;
    c->stream = (hipStream_t)hip_stream;
    if (const char* t = getenv("MCVD_AUTOTUNE")) c->autotune = atoi(t);
    if (const char* t = getenv("MCVD_SIDE_STREAM")) c->side_stream = atoi(t);
    if (const char* t = getenv("MCVD_WINOGRAD")) c->winograd = atoi(t);
    if (const char* t = getenv("MCVD_CONV_DMA1")) c->conv_dma1 = atoi(t);
    if (const char* t = getenv("MCVD_BF16X3")) c->bf16x3 = atoi(t);
    if (const char* t = getenv("MCVD_F16X2")) c->f16x2 = atoi(t);
    if (const char* t = getenv("MCVD_GRAPH")) c->graph = atoi(t);
    if (const char* t = getenv("MCVD_GN_STATS")) c->gn_stats = atoi(t);
    if (const char* t = getenv("MCVD_SPADE_FUSE")) c->spade_fuse = atoi(t);
    const char* e = getenv("MCVD_NAIVE");
    if (e) {
        const int v = atoi(e);
        c->naive_conv = v & 1;
        c->naive_attn = (v >> 1) & 1;
    }
    ctx_register(c);
    *out = c;
    return 0;
    API_CATCH
}

int mcvd_ctx_device_shared(mcvd_ctx* ctx) { return mcvd_ctx_shares_device(ctx) ? 1 : 0; }

void mcvd_ctx_destroy(mcvd_ctx* ctx) {
    if (!ctx) return;
    ctx_unregister(ctx);
    device_lock_release(ctx->device);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->range_flag) (void)hipFree(ctx->range_flag);
    if (ctx->side) (void)hipStreamDestroy(ctx->side);
    if (ctx->cap) (void)hipStreamDestroy(ctx->cap);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    delete ctx;
}

int mcvd_ctx_set_stream(mcvd_ctx* ctx, void* hip_stream) {
    MCVD_REQUIRE(ctx, "ctx is NULL");
    if (ctx->stream != (hipStream_t)hip_stream) {
        std::lock_guard<std::mutex> g(g_dev_mu);       // (mcvd_ctx_shares_device reads the streams of the device's live contexts)
        ctx->stream = (hipStream_t)hip_stream;
        if (ctx->device >= 0 && ctx->device < 64)
            for (mcvd_ctx* o : g_live_ctx[ctx->device]) ++o->epoch;
    }
    return 0;
}

int mcvd_ctx_set_debug_buffer(mcvd_ctx* ctx, void* device_u64) {
    MCVD_REQUIRE(ctx, "ctx is NULL");
    ctx->dbg = (unsigned long long*)device_u64;
    return 0;
}

int mcvd_ctx_set_spade_inputs(mcvd_ctx* ctx, const float* gb, const float* coef2) {
    MCVD_REQUIRE(ctx, "ctx is NULL");
    ctx->spade_gb = gb;
    ctx->spade_coef2 = coef2;
    return 0;
}

int mcvd_ctx_set_stats_buffer(mcvd_ctx* ctx, float* device_floats) {
    MCVD_REQUIRE(ctx, "ctx is NULL");
    ctx->stats_buf = device_floats;
    return 0;
}

int mcvd_ctx_check_range(mcvd_ctx* ctx) {
    API_TRY
    MCVD_REQUIRE(ctx, "ctx is NULL");
    if (!ctx->range_flag) return 0;                // no forward has run under f16x2
    int hit = 0;
    MCVD_HIP_CHECK(hipMemcpyAsync(&hit, ctx->range_flag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    MCVD_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (!hit) return 0;
    MCVD_HIP_CHECK(hipMemsetAsync(ctx->range_flag, 0, sizeof(int), ctx->stream));
    set_error("f16x2: a UNet forward produced non-finite values -- an activation left the fp16 range of the two-piece kernels, or the model "
              "diverged; rerun with the option f16x2 = 0 (the default three-piece bf16 arithmetic has the fp32 range)");
    return MCVD_ERANGE;
    API_CATCH
}

int mcvd_ctx_selftest(mcvd_ctx* ctx) {
    API_TRY
    MCVD_REQUIRE(ctx, "ctx is NULL");
    if (ctx->wino_selftest == 1) return 1;
    if (ctx->wino_selftest < 0) return MCVD_ESELFTEST;
    constexpr int B = 1, Cin = 64, Cout = 96, H = 16;
    const size_t nx = (size_t)B * Cin * H * H, nw = (size_t)Cout * Cin * 9, ny = (size_t)B * Cout * H * H;
    std::vector<float> host(nx + nw + Cout + 2 * Cin + ny);
    unsigned st = 0x9e3779b9u;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return (float)((st >> 8) & 0xffff) / 32768.0f - 1.0f; };     // [-1, 1)
    for (size_t i = 0; i < nx; ++i) host[i] = 1.7f * rnd();
    for (size_t i = 0; i < nw; ++i) host[nx + i] = rnd() * 0.0417f;                       // ~ 1 / sqrt(Cin * 9)
    for (int i = 0; i < Cout; ++i) host[nx + nw + i] = 0.1f * rnd();
    for (int i = 0; i < Cin; ++i) { host[nx + nw + Cout + 2 * i] = 1.0f + 0.3f * rnd(); host[nx + nw + Cout + 2 * i + 1] = 0.3f * rnd(); }
    for (size_t i = 0; i < ny; ++i) host[nx + nw + Cout + 2 * Cin + i] = rnd();
    float* dev = nullptr;
    MCVD_HIP_CHECK(hipMalloc((void**)&dev, (host.size() + 3 * ny) * sizeof(float)));
    struct Free { float* p; ~Free() { (void)hipFree(p); } } guard{dev};
    MCVD_HIP_CHECK(hipMemcpyAsync(dev, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    const float *x = dev, *w = dev + nx, *bias = w + nw, *coef = bias + Cout, *res = coef + 2 * Cin;
    float* y = dev + host.size();
    // The test runs under ITS OWN options: whatever the caller left set on the context (a forced kernel, a statistics buffer the
    // epilogue would write 96-channel partials into, SPADE maps, a debug buffer, the f16x2 arithmetic) is put aside and restored.
    // A launch that cannot run (hipMalloc, a launch error) is NOT a verdict: the state stays 0 ("not yet run"), bf16x3 stays as it
    // was and the real error is returned; only a test that ran and mismatched switches the option off (MCVD_ESELFTEST).
    struct Saved {
        mcvd_ctx* c;
        int conv_shape, naive_conv, f16x2, bf16x3, spade_fuse;
        float* stats_buf;
        const float *spade_gb, *spade_coef2;
        unsigned long long* dbg;
        explicit Saved(mcvd_ctx* c_) : c(c_), conv_shape(c_->conv_shape), naive_conv(c_->naive_conv), f16x2(c_->f16x2), bf16x3(c_->bf16x3),
                                       spade_fuse(c_->spade_fuse), stats_buf(c_->stats_buf), spade_gb(c_->spade_gb), spade_coef2(c_->spade_coef2),
                                       dbg(c_->dbg) {
            c->naive_conv = 0; c->f16x2 = 0; c->bf16x3 = 1; c->spade_fuse = 0;
            c->stats_buf = nullptr; c->spade_gb = nullptr; c->spade_coef2 = nullptr; c->dbg = nullptr;
        }
        ~Saved() {
            c->conv_shape = conv_shape; c->naive_conv = naive_conv; c->f16x2 = f16x2; c->spade_fuse = spade_fuse;
            if (c->wino_selftest >= 0) c->bf16x3 = bf16x3;            // a failed verdict keeps the option off
            c->stats_buf = stats_buf; c->spade_gb = spade_gb; c->spade_coef2 = spade_coef2; c->dbg = dbg;
        }
    };
    const int shapes[3] = {4, 10, 16};
    int rc = 0, ran[3] = {-1, -1, -1};
    std::vector<float> out(3 * ny);
    {
        Saved saved(ctx);
        for (int k = 0; k < 3 && rc == 0; ++k) {
            ctx->conv_shape = shapes[k];
            rc = mcvd_op_conv2d(ctx, x, Cin, nullptr, 0, w, bias, Cout, 3, coef, 1, res, 0.70710678f, y + k * ny, B, H, H);
            ran[k] = last_conv_kernel();
        }
    }
    if (rc) return rc;                     // could not run: mcvd_last_error has the launch's own message; state stays 0
    MCVD_HIP_CHECK(hipMemcpyAsync(out.data(), y, 3 * ny * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    MCVD_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    float scale = 0.0f, d10 = 0.0f;
    size_t diff16 = 0;
    for (size_t i = 0; i < ny; ++i) {
        scale = std::max(scale, std::fabs(out[i]));
        d10 = std::max(d10, std::fabs(out[ny + i] - out[i]));
        diff16 += memcmp(&out[2 * ny + i], &out[ny + i], sizeof(float)) != 0;
    }
    const bool ok = ran[0] == 4 && ran[1] == 10 && ran[2] == 16 && scale > 0.1f && d10 <= 1e-4f * scale && diff16 == 0 && std::isfinite(d10);
    if (ok) { ctx->wino_selftest = 1; return 0; }
    ctx->wino_selftest = -1;
    ctx->bf16x3 = 0;
    ++ctx->epoch;
    set_error("self-test of the hand-scheduled Winograd kernels FAILED on this device (kernels that ran: %d %d %d; |bf16x3 - fp32 MFMA| = %.3e at "
              "scale %.3e; %zu values of the persistent form differ from the per-item form): the option bf16x3 was switched off for this context",
              ran[0], ran[1], ran[2], (double)d10, (double)scale, diff16);
    return MCVD_ESELFTEST;
    API_CATCH
}

int mcvd_ctx_clear_range(mcvd_ctx* ctx) {
    API_TRY
    MCVD_REQUIRE(ctx, "ctx is NULL");
    if (ctx->range_flag) MCVD_HIP_CHECK(hipMemsetAsync(ctx->range_flag, 0, sizeof(int), ctx->stream));
    return 0;
    API_CATCH
}

int mcvd_ctx_set_option(mcvd_ctx* ctx, const char* key, int value) {
    MCVD_REQUIRE(ctx && key, "ctx/key is NULL");
    ++ctx->epoch;                  // captured graphs embed the kernels the options select
    if (!strcmp(key, "naive_conv")) ctx->naive_conv = value;
    else if (!strcmp(key, "naive_attn")) ctx->naive_attn = value;
    else if (!strcmp(key, "fir_form")) ctx->fir_form = value;
    else if (!strcmp(key, "dbg_skip_finalize")) ctx->dbg_skip_finalize = value;
    else if (!strcmp(key, "gn_producer")) ctx->gn_producer = value;
    else if (!strcmp(key, "graph")) ctx->graph = value;
    else if (!strcmp(key, "conv_shape")) ctx->conv_shape = value;
    else if (!strcmp(key, "conv_shape1")) ctx->conv_shape1 = value;
    else if (!strcmp(key, "profile")) ctx->profile = value;
    else if (!strcmp(key, "conv_wdma")) ctx->conv_wdma = value;
    else if (!strcmp(key, "autotune")) ctx->autotune = value;
    else if (!strcmp(key, "side_stream")) ctx->side_stream = value;
    else if (!strcmp(key, "share_fence")) ctx->share_fence = value;
    else if (!strcmp(key, "winograd")) ctx->winograd = value;
    else if (!strcmp(key, "conv_dma1")) ctx->conv_dma1 = value;
    else if (!strcmp(key, "bf16x3")) ctx->bf16x3 = value;
    else if (!strcmp(key, "f16x2")) {
        if (ctx->f16x2 != value) (void)mcvd_ctx_clear_range(ctx);        // a verdict of the other arithmetic's forwards is stale
        ctx->f16x2 = value;
    }
    else if (!strcmp(key, "conv_cot")) ctx->conv_cot = value;
    else if (!strcmp(key, "persist_grid")) ctx->persist_grid = value;
    else if (!strcmp(key, "temb_table")) ctx->temb_table = value;
    else if (!strcmp(key, "im2col_lds")) ctx->im2col_lds = value;
    else if (!strcmp(key, "gn_stats")) ctx->gn_stats = value;
    else if (!strcmp(key, "spade_fuse")) ctx->spade_fuse = value;
    else if (!strcmp(key, "attn_presplit")) ctx->attn_presplit = value;
    else {
        set_error("unknown option '%s'", key);
        return MCVD_EINVAL;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------ model
static void default_schedule(mcvd_model* m) {
    // models/__init__.py:24-32 + ncsnpp_more.py:735-743 in fp32.  torch.linspace is symmetric about the midpoint
    // (ATen RangeFactories: start + step*i for i < steps/2, end - step*(steps-1-i) otherwise).
    const int T = m->d.num_classes;
    m->betas.assign(T, 0.f);
    m->alphas.assign(T, 0.f);
    m->alphas_prev.assign(T, 0.f);
    if (m->d.sigma_dist == 0) {
        const float start = m->d.sigma_begin, end = m->d.sigma_end;
        const float step = (end - start) / (float)(T - 1);
        for (int i = 0; i < T; ++i) m->betas[i] = (i < T / 2) ? start + step * (float)i : end - step * (float)(T - 1 - i);
        float p = 1.0f;                      // alphas = cumprod(1 - betas.flip(0)).flip(0)
        for (int i = T - 1; i >= 0; --i) {
            p = p * (1.0f - m->betas[i]);
            m->alphas[i] = p;
        }
        for (int i = 0; i < T; ++i) m->alphas_prev[i] = (i + 1 < T) ? m->alphas[i + 1] : 1.0f;
    } else {
        // cosine: t = linspace(T, 0, T+1)/T ; f = cos((t+s)/(1+s)*pi/2)^2 ; alphas = f[:-1]/f[-1]
        std::vector<float> f(T + 1);
        const float s = 0.008f;
        for (int i = 0; i <= T; ++i) {
            const float t = (float)(T - i) / (float)T;
            const float c = cosf((t + s) / (1.0f + s) * (float)(M_PI / 2));
            f[i] = c * c;
        }
        for (int i = 0; i < T; ++i) m->alphas[i] = f[i] / f[T];
        for (int i = 0; i < T; ++i) m->alphas_prev[i] = (i + 1 < T) ? m->alphas[i + 1] : 1.0f;
        for (int i = 0; i < T; ++i) m->betas[i] = 1.0f - m->alphas[i] / m->alphas_prev[i];
    }
    // layers.py:507-510: exp(arange(half) * -(log(10000)/(half-1))) in fp32
    const int half = m->d.ngf / 2;
    m->freqs.assign(half, 0.f);
    const float e = (float)(-(log(10000.0) / (double)(half - 1)));
    for (int k = 0; k < half; ++k) m->freqs[k] = expf((float)k * e);
}

int mcvd_model_create(mcvd_ctx* ctx, const mcvd_unet_desc* desc, mcvd_model** out) {
    API_TRY
    MCVD_REQUIRE(desc && out, "model_create: NULL argument");
    MCVD_REQUIRE(desc->sigma_dist == 0 || desc->sigma_dist == 1, "desc: sigma_dist=%d", desc->sigma_dist);
    mcvd_model* m = new mcvd_model();
    m->ctx = ctx;
    m->d = *desc;
    if (m->build_plan()) {
        delete m;
        return MCVD_EINVAL;
    }
    default_schedule(m);
    if (ctx) {      // ctx == NULL: plan-only model (parameter table, schedule, launch count) for GPU-less host tests
        hipError_t e = hipMalloc((void**)&m->blob, (size_t)m->blob_floats * sizeof(float));
        if (e != hipSuccess) {
            set_error("hipMalloc(param blob, %lld floats): %s", (long long)m->blob_floats, hipGetErrorString(e));
            delete m;
            return MCVD_EHIP;
        }
        MCVD_HIP_CHECK(hipMemsetAsync(m->blob, 0, (size_t)m->blob_floats * sizeof(float), ctx->stream));
    }
    *out = m;
    return 0;
    API_CATCH
}

void mcvd_model_destroy(mcvd_model* m) {
    if (!m) return;
    if (m->ctx) (void)hipStreamSynchronize(m->ctx->stream);
    m->drop_graph();
    if (m->blob) (void)hipFree(m->blob);
    if (m->packed) (void)hipFree(m->packed);
    if (m->packed_h) (void)hipFree(m->packed_h);
    if (m->coef2_desc_dev) (void)hipFree(m->coef2_desc_dev);
    if (m->arena) (void)hipFree(m->arena);
    if (m->labels) (void)hipFree(m->labels);
    if (m->eps_buf) (void)hipFree(m->eps_buf);
    if (m->ksplit_buf) (void)hipFree(m->ksplit_buf);
    if (m->labels_f) (void)hipFree(m->labels_f);
    if (m->temb_tab) (void)hipFree(m->temb_tab);
    if (m->temb_tmp) (void)hipFree(m->temb_tmp);
    if (m->temb_lab) (void)hipFree(m->temb_lab);
    if (m->fp_buf) (void)hipFree(m->fp_buf);
    if (m->alphas_dev) (void)hipFree(m->alphas_dev);
    if (m->cond_z) (void)hipFree(m->cond_z);
    if (m->noise_buf) (void)hipFree(m->noise_buf);
    for (hipEvent_t e : m->ev) (void)hipEventDestroy(e);
    delete m;
}

int mcvd_model_num_params(mcvd_model* m) { return m ? (int)m->params.size() : MCVD_EINVAL; }

int mcvd_model_param_info(mcvd_model* m, int index, const char** name, int64_t shape[4], int* ndim,
                          int64_t* blob_offset_floats) {
    MCVD_REQUIRE(m && index >= 0 && index < (int)m->params.size(), "param_info: index %d out of range", index);
    const ParamInfo& p = m->params[index];
    if (name) *name = p.name.c_str();
    if (shape) memcpy(shape, p.shape, sizeof(p.shape));
    if (ndim) *ndim = p.ndim;
    if (blob_offset_floats) *blob_offset_floats = p.off;
    return 0;
}

int mcvd_model_set_param(mcvd_model* m, const char* name, const float* data, const int64_t* shape, int ndim,
                         int data_on_device) {
    API_TRY
    MCVD_REQUIRE(m && name && data, "set_param: NULL argument");
    MCVD_REQUIRE(m->ctx, "set_param: plan-only model (created without a ctx)");
    const int i = m->find_param(name);
    MCVD_REQUIRE(i >= 0, "set_param: unknown parameter '%s'", name);
    ParamInfo& p = m->params[i];
    bool same = (ndim == p.ndim);
    for (int k = 0; same && k < ndim; ++k) same = (shape[k] == p.shape[k]);
    MCVD_REQUIRE(same, "set_param: shape mismatch for '%s'", name);
    MCVD_HIP_CHECK(hipMemcpyAsync(m->blob + p.off, data, (size_t)p.numel * sizeof(float),
                                  data_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, m->ctx->stream));
    if (!data_on_device) MCVD_HIP_CHECK(hipStreamSynchronize(m->ctx->stream));   // caller may free the host buffer
    p.set = true;
    m->finalized = false;
    return 0;
    API_CATCH
}

int mcvd_model_blob_floats(mcvd_model* m, int64_t* n_floats) {
    MCVD_REQUIRE(m && n_floats, "blob_floats: NULL argument");
    *n_floats = m->blob_floats;
    return 0;
}

int mcvd_model_export_blob(mcvd_model* m, float* dst_device) {
    MCVD_REQUIRE(m && dst_device && m->ctx, "export_blob: NULL argument");
    MCVD_HIP_CHECK(hipMemcpyAsync(dst_device, m->blob, (size_t)m->blob_floats * sizeof(float), hipMemcpyDeviceToDevice,
                                  m->ctx->stream));
    return 0;
}

int mcvd_model_import_blob(mcvd_model* m, const float* src_device) {
    MCVD_REQUIRE(m && src_device && m->ctx, "import_blob: NULL argument");
    MCVD_HIP_CHECK(hipMemcpyAsync(m->blob, src_device, (size_t)m->blob_floats * sizeof(float), hipMemcpyDeviceToDevice,
                                  m->ctx->stream));
    for (auto& p : m->params) p.set = true;
    m->finalized = false;
    return 0;
}

// RCCL is resolved at run time: libmcvd_hip.so links the HIP runtime only (single-GPU users never load librccl)
int mcvd_model_broadcast_params(mcvd_model* m, void* rccl_comm, int root) {
    API_TRY
    MCVD_REQUIRE(m && m->ctx && rccl_comm && root >= 0, "broadcast_params: NULL model / communicator or negative root");
    typedef int (*bcast_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
    typedef const char* (*errstr_fn)(int);
    static bcast_fn bcast = nullptr;
    static errstr_fn errstr = nullptr;
    if (!bcast) {
        void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        MCVD_REQUIRE(h, "broadcast_params: librccl.so not found (%s)", dlerror());
        bcast = reinterpret_cast<bcast_fn>(dlsym(h, "ncclBroadcast"));
        errstr = reinterpret_cast<errstr_fn>(dlsym(h, "ncclGetErrorString"));
        MCVD_REQUIRE(bcast, "broadcast_params: ncclBroadcast not exported by librccl");
    }
    const int ncclFloat32 = 7;
    const int rc = bcast(m->blob, m->blob, (size_t)m->blob_floats, ncclFloat32, root, rccl_comm, m->ctx->stream);
    MCVD_REQUIRE(rc == 0, "broadcast_params: ncclBroadcast failed: %s", errstr ? errstr(rc) : "?");
    for (auto& p : m->params) p.set = true;
    m->finalized = false;
    return 0;
    API_CATCH
}

int mcvd_model_finalize(mcvd_model* m) {
    API_TRY
    MCVD_REQUIRE(m && m->ctx, "finalize: NULL model or plan-only model");
    for (const auto& p : m->params) MCVD_REQUIRE(p.set, "finalize: parameter '%s' was never set", p.name.c_str());
    hipStream_t s = m->ctx->stream;
    if (!m->packed) MCVD_HIP_CHECK(hipMalloc((void**)&m->packed, (size_t)m->packed_floats * sizeof(float)));
    MCVD_HIP_CHECK(hipMemsetAsync(m->packed, 0, (size_t)m->packed_floats * sizeof(float), s));
    for (const ConvPack& p : m->packs) {
        for (size_t j = 0; j < p.weights.size(); ++j) {
            const ParamInfo& w = m->params[m->find_param(p.weights[j].c_str())];
            const ParamInfo& b = m->params[m->find_param(p.biases[j].c_str())];
            if (int rc = launch_pack_conv_weight(m->blob + w.off, m->packed + p.wp, p.Cout_each, p.Cin, p.ks, p.CinP, p.CoutP,
                                                 p.nin, (int)j * p.Cout_each, s))
                return rc;
            if (!p.zero_bias)
                MCVD_HIP_CHECK(hipMemcpyAsync(m->packed + p.bias + j * p.Cout_each, m->blob + b.off,
                                              (size_t)p.Cout_each * sizeof(float), hipMemcpyDeviceToDevice, s));
            if (!p.extra_bias.empty()) {          // packed bias += the up-block shortcut's bias (model.cpp: res_block)
                const ParamInfo& e = m->params[m->find_param(p.extra_bias.c_str())];
                const float* in[4] = {m->packed + p.bias, m->blob + e.off, nullptr, nullptr};
                const float w2[4] = {1.0f, 1.0f, 0.0f, 0.0f};
                if (int rc = launch_lincomb(m->packed + p.bias, in, w2, 1.0f, 2, p.Cout_each, s)) return rc;
            }
            if (p.wpw >= 0)
                if (int rc = launch_pack_wino_weight(m->blob + w.off, m->packed + p.wpw, p.Cout_each, p.Cin, p.CinP, p.CoutP, s)) return rc;
            if (p.wpb >= 0 && p.ks == 3)
                if (int rc = launch_pack_wino3_weight(m->blob + w.off, m->packed + p.wpb, p.Cout_each, p.Cin, p.CinP, p.CoutP, s)) return rc;
            if (p.alt_kind) {              // the conv's GEMM form: its fp32 matrix, then the three bf16 pieces of it (kernels/conv_gemm_forms.cpp)
                if (int rc = launch_pack_conv_gemm_form(m->blob + w.off, m->packed + p.alt_wp, p.Cout_each, p.Cin, p.alt_kind, p.alt_CoutP, s)) return rc;
                if (int rc = launch_pack_conv1x1_h2(m->packed + p.alt_wp, m->packed + p.alt_wpb, p.alt_CinP, p.alt_CoutP, s, 3)) return rc;
            }
        }
        if (p.wpb >= 0 && p.ks == 1)           // from the packed fp32 matrix: every fused weight (q | k | v) and the padding are in place
            if (int rc = launch_pack_conv1x1_h2(m->packed + p.wp, m->packed + p.wpb, p.CinP, p.CoutP, s, 3)) return rc;
    }
    for (const DenseEntry& e : m->dense) {
        const ParamInfo& w = m->params[m->find_param(e.weight.c_str())];
        const ParamInfo& b = m->params[m->find_param(e.bias.c_str())];
        if (int rc = launch_transpose_into(m->blob + w.off, m->packed + m->dense_wt, 2 * e.ch, m->T, m->NE, e.emb_off, s)) return rc;
        MCVD_HIP_CHECK(hipMemcpyAsync(m->packed + m->dense_bias + e.emb_off, m->blob + b.off, (size_t)2 * e.ch * sizeof(float),
                                      hipMemcpyDeviceToDevice, s));
    }
    MCVD_HIP_CHECK(hipMemcpyAsync(m->packed + m->freqs_off, m->freqs.data(), m->freqs.size() * sizeof(float),
                                  hipMemcpyHostToDevice, s));
    {   // SPADE nets: the descriptors of every (1 + scale, shift) table (model.cpp: OP_COEF2)
        std::vector<long long> desc;
        m->coef2_first = -1; m->coef2_count = 0; m->coef2_cmax = 0;
        bool all_arena = true;
        for (size_t i = 0; i < m->ops.size(); ++i) {
            const Op& op = m->ops[i];
            if (op.kind != OP_COEF2 || op.prep) continue;
            if (m->coef2_first < 0) m->coef2_first = (int)i;
            all_arena = all_arena && op.dst.kind == REF_ARENA;
            desc.push_back((long long)op.dst.off); desc.push_back(op.emb_off); desc.push_back(op.Cout);
            m->coef2_cmax = std::max(m->coef2_cmax, op.Cout);
            ++m->coef2_count;
        }
        if (m->coef2_count > 1 && all_arena) {
            if (m->coef2_desc_dev) (void)hipFree(m->coef2_desc_dev);
            m->coef2_desc_dev = nullptr;
            MCVD_HIP_CHECK(hipMalloc((void**)&m->coef2_desc_dev, desc.size() * sizeof(long long)));
            MCVD_HIP_CHECK(hipMemcpyAsync(m->coef2_desc_dev, desc.data(), desc.size() * sizeof(long long), hipMemcpyHostToDevice, s));
            MCVD_HIP_CHECK(hipStreamSynchronize(s));          // `desc` is a local
        } else {
            m->coef2_count = 0;
        }
    }
    MCVD_HIP_CHECK(hipStreamSynchronize(s));
    m->finalized = true;
    m->packed_h_valid = false;             // the two-piece fp16 forms follow the weights: repacked on the next use under f16x2
    ++m->epoch;
    if (m->ctx->f16x2)
        if (int rc = m->ensure_f16x2_weights()) return rc;
    // once per context: the hand-scheduled Winograd kernels against the compiler-scheduled one, on this device (a failure switches the
    // option bf16x3 off and is reported by mcvd_ctx_selftest / mcvd_last_error; the model stays usable on the fp32-MFMA kernels)
    if (m->ctx->wino_selftest == 0 && m->ctx->bf16x3 && m->ctx->winograd) (void)mcvd_ctx_selftest(m->ctx);
    return 0;
    API_CATCH
}

int mcvd_model_get_schedule(mcvd_model* m, float* betas, float* alphas, float* alphas_prev, int n) {
    MCVD_REQUIRE(m && n == m->d.num_classes, "get_schedule: n must equal num_classes");
    if (betas) memcpy(betas, m->betas.data(), n * sizeof(float));
    if (alphas) memcpy(alphas, m->alphas.data(), n * sizeof(float));
    if (alphas_prev) memcpy(alphas_prev, m->alphas_prev.data(), n * sizeof(float));
    return 0;
}

int mcvd_model_set_schedule(mcvd_model* m, const float* betas, const float* alphas, const float* alphas_prev, int n) {
    MCVD_REQUIRE(m && betas && alphas && alphas_prev && n == m->d.num_classes, "set_schedule: bad arguments");
    m->betas.assign(betas, betas + n);
    m->alphas.assign(alphas, alphas + n);
    m->alphas_prev.assign(alphas_prev, alphas_prev + n);
    m->alphas_dev_valid = false;
    return 0;
}

int mcvd_model_set_gamma_tables(mcvd_model* m, const float* k_cum, const float* theta_t, int n) {
    MCVD_REQUIRE(m && k_cum && theta_t && n == m->d.num_classes, "set_gamma_tables: bad arguments");
    m->k_cum.assign(k_cum, k_cum + n);
    m->theta_t.assign(theta_t, theta_t + n);
    return 0;
}

int mcvd_model_set_cond_noise(mcvd_model* m, const float* z_device, uint64_t seed, uint64_t sample_offset, uint64_t first_draw) {
    MCVD_REQUIRE(m, "set_cond_noise: NULL model");
    m->cond_noise_src = z_device;
    m->cond_noise_seed = seed;
    m->cond_noise_offset = sample_offset;
    m->cond_noise_draw = first_draw;
    return 0;
}

int mcvd_model_set_temb_freqs(mcvd_model* m, const float* freqs_host, int n) {
    MCVD_REQUIRE(m && freqs_host && n == m->d.ngf / 2, "set_temb_freqs: n must equal ngf/2");
    m->freqs.assign(freqs_host, freqs_host + n);
    if (m->packed && m->ctx) {
        MCVD_HIP_CHECK(hipMemcpyAsync(m->packed + m->freqs_off, m->freqs.data(), n * sizeof(float), hipMemcpyHostToDevice,
                                      m->ctx->stream));
        MCVD_HIP_CHECK(hipStreamSynchronize(m->ctx->stream));
    }
    return 0;
}

int mcvd_unet_forward(mcvd_model* m, const float* x, const int64_t* labels, const float* cond, float* eps_out, int B) {
    API_TRY
    MCVD_REQUIRE(m, "forward: NULL model");
    return m->forward(x, labels, cond, eps_out, B);
    API_CATCH
}

int mcvd_unet_forward_masked(mcvd_model* m, const float* x, const int64_t* labels, const float* cond, const int32_t* cond_mask,
                             float* eps_out, int B) {
    API_TRY
    MCVD_REQUIRE(m, "forward: NULL model");
    m->cond_mask = cond_mask;
    const int rc = m->forward(x, labels, cond, eps_out, B);
    m->cond_mask = nullptr;
    return rc;
    API_CATCH
}

int mcvd_unet_forward_ft(mcvd_model* m, const float* x, const float* t, const float* cond, float* eps_out, int B) {
    API_TRY
    MCVD_REQUIRE(m, "forward: NULL model");
    m->labels_f32 = 1;
    const int rc = m->forward(x, t, cond, eps_out, B);
    m->labels_f32 = 0;
    return rc;
    API_CATCH
}

int mcvd_lincomb(mcvd_ctx* ctx, float* out, const float* in0, const float* in1, const float* in2, const float* in3, float w0,
                 float w1, float w2, float w3, float scale, int nin, int64_t n) {
    MCVD_REQUIRE(ctx && out && in0 && nin >= 1 && nin <= 4 && n >= 0, "lincomb: bad arguments");
    const float* in[4] = {in0, in1, in2, in3};
    const float w[4] = {w0, w1, w2, w3};
    for (int k = 0; k < nin; ++k) MCVD_REQUIRE(in[k], "lincomb: input %d is NULL", k);
    return launch_lincomb(out, in, w, scale, nin, n, ctx->stream);
}

int mcvd_pndm_transfer(mcvd_ctx* ctx, float* out, const float* x, const float* e, float d, float c1, float c2, int clip,
                       int64_t n) {
    MCVD_REQUIRE(ctx && out && x && e && n >= 0, "pndm_transfer: bad arguments");
    return launch_pndm_transfer(out, x, e, d, c1, c2, clip, n, ctx->stream);
}

int mcvd_model_num_launches(mcvd_model* m, int) {
    if (!m) return MCVD_EINVAL;
    int n = 0;
    for (const Op& op : m->ops) n += op.prep ? 0 : 1;
    return n;
}

int mcvd_model_get_tuning(mcvd_model* m, int B, int* shapes, int* cots, int cap) {
    MCVD_REQUIRE(m && B > 0, "get_tuning: bad arguments");
    auto it = m->tuned_cache.find(B);
    MCVD_REQUIRE(it != m->tuned_cache.end(), "get_tuning: batch size %d has not been tuned", B);
    const int n = (int)it->second.first.size();
    if (!shapes) return n;
    MCVD_REQUIRE(cots && cap >= n, "get_tuning: capacity %d < %d ops", cap, n);
    for (int i = 0; i < n; ++i) {
        shapes[i] = it->second.first[i];
        cots[i] = it->second.second[i];
    }
    return n;
}

int mcvd_model_set_tuning(mcvd_model* m, int B, const int* shapes, const int* cots, int n) {
    API_TRY
    MCVD_REQUIRE(m && shapes && cots && B > 0, "set_tuning: bad arguments");
    MCVD_REQUIRE(n == (int)m->ops.size(), "set_tuning: %d entries for a plan of %d ops (tuning of another model?)", n, (int)m->ops.size());
    for (int i = 0; i < n; ++i) {
        const bool conv = m->ops[i].kind == OP_CONV;
        MCVD_REQUIRE(conv ? (((shapes[i] >= -1 && shapes[i] <= 23) || shapes[i] == 36 || shapes[i] == 40) && cots[i] >= 0 && cots[i] <= 9) : shapes[i] == -1,
                     "set_tuning: entry %d (shape %d, cout tile %d) does not fit op kind %d", i, shapes[i], cots[i], (int)m->ops[i].kind);
    }
    if (m->ctx) m->sync_tuning_options();         // the table belongs to the options in force now; a later option change drops it
    m->tuned_cache[B] = {std::vector<int>(shapes, shapes + n), std::vector<int>(cots, cots + n)};
    if (m->tuned_B == B) m->tuned_B = 0;              // re-read on the next forward
    return 0;
    API_CATCH
}

int mcvd_model_prepare_cond(mcvd_model* m, const float* cond, int B) {
    API_TRY
    MCVD_REQUIRE(m && m->ctx, "prepare_cond: NULL model");
    return m->prepare_cond(cond, B);
    API_CATCH
}

int mcvd_model_invalidate_cond(mcvd_model* m) {
    MCVD_REQUIRE(m, "invalidate_cond: NULL model");
    m->cond_cache_valid = false;
    return 0;
}

// Per-op timings of the last event-instrumented forward (option "profile").  Arrays of length >= n_ops:
// kind (OpKind), ks (conv kernel size or 0), ms, algorithmic flops, algorithmic bytes (inputs + outputs + weights once).
int mcvd_model_profile_read(mcvd_model* m, int* kinds, int* ks, double* ms, double* flops, double* bytes, int cap) {
    MCVD_REQUIRE(m && m->ctx, "profile_read: NULL model");
    const int n = (int)m->ops.size();
    if (!kinds) return n;
    MCVD_REQUIRE(cap >= n, "profile_read: capacity %d < %d ops", cap, n);
    MCVD_REQUIRE(m->ev.size() == 2 * (size_t)n && m->profile_B > 0, "profile_read: no instrumented forward recorded");
    MCVD_HIP_CHECK(hipStreamSynchronize(m->ctx->stream));
    const double B = m->profile_B;
    for (int i = 0; i < n; ++i) {
        const Op& op = m->ops[i];
        float t = 0.f;
        // (the (1 + scale, shift) table ops behind the first one launch nothing: the first fills every table -- model.cpp OP_COEF2)
        // ... and in a device-loop sampler call with the embedding table live the time MLP and the Dense_0 projections ran once, before the loop
        const bool noop = (op.kind == OP_COEF2 && m->coef2_count > 1 && (int)i != m->coef2_first) ||
                          (m->profile_temb_skipped && (op.kind == OP_TEMB || op.kind == OP_DENSE));
        if (!op.prep && !noop) MCVD_HIP_CHECK(hipEventElapsedTime(&t, m->ev[2 * i], m->ev[2 * i + 1]));
        kinds[i] = (int)op.kind;
        ks[i] = op.kind == OP_CONV ? op.ks : 0;
        ms[i] = t;
        const double HW = (double)op.H * op.W;
        const double cin = op.src0.C + (op.src1.kind == REF_NONE ? 0 : op.src1.C);
        double f = 0, by = 0;
        switch (op.kind) {
            case OP_CONV:
                f = 2.0 * B * HW * op.Cout * cin * op.ks * op.ks;
                by = 4.0 * (B * HW * (cin + op.Cout + (op.res.kind != REF_NONE ? op.Cout : 0)) + cin * op.ks * op.ks * op.Cout);
                break;
            case OP_GN: by = 4.0 * B * HW * cin; f = 4.0 * B * HW * cin; break;       // one algorithmic read (2nd pass hits L2)
            case OP_FIR: {
                const double o = op.up ? 4.0 : 0.25;
                by = 4.0 * B * HW * op.src0.C * (1.0 + o * (op.dst2.kind != REF_NONE ? 2.0 : 1.0));
                f = B * HW * op.src0.C * o * (op.up ? 8.0 : 32.0);
                break;
            }
            case OP_ATTN:
                f = 4.0 * B * HW * HW * op.Cout;                                     // QK^T + PV
                by = 4.0 * B * HW * op.Cout * 4.0;                                    // q,k,v in + o out
                break;
            case OP_APPLY: by = 4.0 * B * HW * cin * 4.0; f = 12.0 * B * HW * cin; break;   // x, gamma, beta in; y out
            case OP_TEMB: f = 2.0 * B * (m->d.ngf * m->T + (double)m->T * m->T); by = 4.0 * (m->d.ngf * m->T + (double)m->T * m->T); break;
            case OP_DENSE: f = 2.0 * B * m->T * m->NE; by = 4.0 * ((double)m->T * m->NE + B * m->NE); break;
            default: break;
        }
        flops[i] = f;
        bytes[i] = by;
    }
    return n;
}

// info: kind, module, ks, H, Cin, Cout, has_res | tuned_shape<<4 | tuned_cot<<8, has_coef
int mcvd_model_op_info(mcvd_model* m, int i, int info[8]) {
    MCVD_REQUIRE(m && info && i >= 0 && i < (int)m->ops.size(), "op_info: index %d", i);
    const Op& op = m->ops[i];
    info[0] = (int)op.kind; info[1] = op.module; info[2] = op.ks; info[3] = op.H;
    info[4] = op.src0.C + (op.src1.kind == REF_NONE ? 0 : op.src1.C); info[5] = op.Cout;
    info[6] = op.res.kind != REF_NONE; info[7] = op.coef.kind != REF_NONE;
    if ((size_t)i < m->tuned_shape.size() && m->tuned_shape[i] >= 0) info[6] |= (m->tuned_shape[i] << 4) | (m->tuned_cot[i] << 8) | (1 << 12);
    return 0;
}

int mcvd_model_op_kernel(mcvd_model* m, int i) {
    if (!m || i < 0 || i >= (int)m->ops.size() || m->ops[i].kind != OP_CONV || (size_t)i >= m->ran_kernel.size()) return -1;
    return m->ran_kernel[i];
}

long mcvd_model_fused_launches(mcvd_model* m, int what) { return (m && what >= 0 && what < 4) ? m->fused_launches[what] : -1; }

int mcvd_model_module_output(mcvd_model* m, int module, int B, float* dst, int64_t capacity, int* C, int* H) {
    MCVD_REQUIRE(m && dst && B > 0 && B <= m->arena_B, "module_output: run a forward at batch >= B first");
    const Op* last = nullptr;
    for (const Op& op : m->ops)
        if (op.module == module && op.dst.kind == REF_ARENA) last = &op;
    MCVD_REQUIRE(last, "module_output: module %d has no workspace output", module);
    const int c = last->dst.C;
    int h = last->H;
    if (last->kind == OP_FIR) h = last->up ? 2 * h : h / 2;
    if (last->kind == OP_TEMB || last->kind == OP_DENSE) h = 0;
    const int64_t per = h ? (int64_t)c * h * h : c;
    MCVD_REQUIRE(per * B <= capacity, "module_output: capacity %lld < %lld", (long long)capacity, (long long)(per * B));
    MCVD_HIP_CHECK(hipMemcpyAsync(dst, m->arena + last->dst.off * (int64_t)B, (size_t)per * B * sizeof(float),
                                  hipMemcpyDeviceToDevice, m->ctx->stream));
    if (C) *C = c;
    if (H) *H = h;
    return 0;
}

// ------------------------------------------------------------------------------------------------ sampler
int mcvd_sampler_run(mcvd_model* m, int kind, float* x, const float* cond, const float* noise, uint64_t seed,
                     uint64_t sample_offset, int subsample_steps, int flags, double t_min, int B) {
    API_TRY
    MCVD_REQUIRE(m && x && B > 0, "sampler_run: bad arguments");
    MCVD_REQUIRE(kind == MCVD_SAMPLER_DDPM || kind == MCVD_SAMPLER_DDIM, "sampler_run: kind %d", kind);
    MCVD_REQUIRE(m->finalized, "sampler_run before mcvd_model_finalize");
    if (int rc = mcvd_ctx_clear_range(m->ctx)) return rc;          // the verdict at the end is about THIS call's forwards
    const int T = m->d.num_classes;
    // schedule subsampling, models/__init__.py:229-237
    std::vector<int> steps;
    std::vector<float> al, alp, be;
    const bool subsampled = subsample_steps > 0 && subsample_steps < T;
    if (subsampled) {
        const int skip = T / subsample_steps;
        for (int t = 0; t < T; t += skip) steps.push_back(t);
        const int L = (int)steps.size();
        al.resize(L); alp.resize(L); be.resize(L);
        for (int i = 0; i < L; ++i) al[i] = m->alphas[steps[i]];
        for (int i = 0; i < L; ++i) alp[i] = (i + 1 < L) ? al[i + 1] : 1.0f;
        for (int i = 0; i < L; ++i) be[i] = 1.0f - al[i] / alp[i];
    } else {
        for (int t = 0; t < T; ++t) steps.push_back(t);
        al = m->alphas; alp = m->alphas_prev; be = m->betas;
    }
    const int L = (int)steps.size();
    const int64_t per = (int64_t)m->d.channels * m->d.num_frames * m->d.image_size * m->d.image_size;
    const int64_t n = per * B;
    if (int rc = m->prepare_B(B)) return rc;
    if (int rc = m->prepare_cond(cond, B)) return rc;      // SPADE: gamma/beta once per sampler call (cond is constant)
    struct CacheGuard { mcvd_model* m; ~CacheGuard() { m->cond_cache_valid = false; } } guard{m};
    hipStream_t s = m->ctx->stream;
    m->profile_armed = true;          // with option "profile": the first forward of this call is event-instrumented
    const bool gam = (flags & MCVD_FLAG_GAMMA) != 0;
    if (gam) {
        MCVD_REQUIRE(kind == MCVD_SAMPLER_DDPM, "sampler_run: gamma noise exists for the DDPM sampler only");
        MCVD_REQUIRE(m->d.gamma && (int)m->k_cum.size() == T && m->noise_buf, "sampler_run: MCVD_FLAG_GAMMA needs a model.gamma net "
                     "with mcvd_model_set_gamma_tables");
    }
    const bool cond_lib_noise = m->d.noise_in_cond && m->d.num_frames_cond > 0 && !m->cond_noise_src;
    if (cond_lib_noise) {             // conditioning noise from the library's own stream for this call
        m->cond_noise_seed = seed;
        m->cond_noise_offset = sample_offset;
        m->cond_noise_draw = 0;
    }
    struct CondGammaGuard { mcvd_model* m; ~CondGammaGuard() { m->cond_gamma_k = 0.f; m->uniform_labels = 0; m->temb_row_live = 0; } } cg_guard{m};
    m->uniform_labels = 1;            // every forward of the loop labels all rows alike (:283, :332)
    // the labels of every forward of this call, in order: the executed steps (:269-270, :283), then L - 1 for the denoise pass (:332)
    int fwd_no = 0;
    if (m->ctx->temb_table && !m->d.cond_emb) {
        std::vector<int> fl;
        for (int i = 0; i < L; ++i) {
            const double thr = t_min * (double)L;
            if (subsampled ? ((float)steps[i] < (float)thr) : ((double)steps[i] < thr)) continue;
            fl.push_back(steps[i]);
        }
        if (flags & MCVD_FLAG_DENOISE) fl.push_back(L - 1);
        if (!fl.empty())
            if (int rc = m->prepare_temb_table(fl)) return rc;
    }
    auto set_cond_gamma = [&](int label) {         // ncsnpp_more.py:761-765: k_cum[labels], theta_t[labels], alphas[labels]
        // (the NETWORK decides by its own config -- model.gamma -- whatever `gamma` the sampler was called with: a DDIM call, or gamma=False on a gamma net)
        if (!(m->d.gamma && m->d.noise_in_cond && (int)m->k_cum.size() == T)) return;
        m->cond_gamma_k = m->k_cum[label];
        m->cond_gamma_theta = m->theta_t[label];
        m->cond_gamma_kt = m->k_cum[label] * m->theta_t[label];
        m->cond_gamma_sd = sqrtf(1.0f - m->alphas[label]);
    };
    const int use_philox = (noise || gam) ? 0 : 1;
    uint64_t draw = 0;
    bool started = false;
    // gamma: z_i = (g - k_i theta_i) / sqrt(1 - a_i), g ~ Gamma(k_i, scale theta_i) or the injected raw draw  (:273-276, :319-322)
    auto gamma_draw = [&](int i) -> int {
        const float k = m->k_cum[steps[i]], th = m->theta_t[steps[i]];
        return launch_gamma_noise(m->noise_buf, noise ? noise + draw * n : nullptr, k, th, k * th, sqrtf(1.0f - al[i]), seed, sample_offset,
                                  draw, B, per, s);
    };
    for (int i = 0; i < L; ++i) {
        // :269-270 `step < t_min*len(alphas)`: a 0-dim int64 tensor against a Python float on the subsampled schedule (torch compares in
        // float32), a numpy int64 against it otherwise (double)
        const double thr = t_min * (double)L;
        if (subsampled ? ((float)steps[i] < (float)thr) : ((double)steps[i] < thr)) continue;
        const float a = al[i], ap = alp[i], b = be[i];
        if (!started && t_min > 0.0) {                                                    // :272-279
            if (gam) {
                if (int rc = gamma_draw(i)) return rc;
            }
            if (int rc = launch_renoise(x, gam ? m->noise_buf : (noise ? noise + draw * n : nullptr), sqrtf(a), sqrtf(1.0f - a), n,
                                        use_philox, seed, sample_offset, draw, per, s))
                return rc;
            ++draw;
        }
        started = true;
        // :283 (with the call's embedding table live nothing but a noise_in_cond net reads the labels: the fill is skipped)
        const bool need_labels = !m->temb_row_live || m->d.noise_in_cond;
        if (need_labels)
            if (int rc = launch_fill_labels(m->labels, steps[i], B, s)) return rc;
        set_cond_gamma(steps[i]);
        if (m->temb_row_live)
            if (int rc = m->use_temb_row(fwd_no++, B)) return rc;
        if (int rc = m->forward(x, m->labels, cond, m->eps_buf, B)) return rc;           // :284
        const float c_x0a = 1.0f / sqrtf(a), c_x0b = sqrtf(1.0f - a);                    // :287
        float c0, c1, cn = 0.0f;
        if (kind == MCVD_SAMPLER_DDPM) {
            c0 = sqrtf(ap) * b / (1.0f - a);                                             // :290
            c1 = sqrtf(1.0f - b) * (1.0f - ap) / (1.0f - a);
            if (i + 1 != L)                                                              // :311-328
                cn = (flags & MCVD_FLAG_JUST_BETA) ? sqrtf(b) : sqrtf((1.0f - ap) / (1.0f - a) * b);
        } else {
            c0 = sqrtf(ap);                                                              // :168
            c1 = sqrtf(1.0f - ap);
        }
        const bool draws = (kind == MCVD_SAMPLER_DDPM) && (i + 1 != L);
        if (draws && gam)
            if (int rc = gamma_draw(i)) return rc;
        const float* zsrc = !draws ? nullptr : (gam ? m->noise_buf : (noise ? noise + draw * n : nullptr));
        if (int rc = launch_sampler_update(kind, x, m->eps_buf, zsrc, c_x0a, c_x0b, c0,
                                           c1, cn, (flags & MCVD_FLAG_CLIP_BEFORE) ? 1 : 0, n, (draws && use_philox) ? 1 : 0,
                                           seed, sample_offset, draw, per, s))
            return rc;
        if (draws) ++draw;
    }
    if (flags & MCVD_FLAG_DENOISE) {                                                      // :331-333, label L-1 (sic)
        if (!m->temb_row_live || m->d.noise_in_cond)
            if (int rc = launch_fill_labels(m->labels, L - 1, B, s)) return rc;
        set_cond_gamma(L - 1);
        if (m->temb_row_live)
            if (int rc = m->use_temb_row(fwd_no++, B)) return rc;
        if (int rc = m->forward(x, m->labels, cond, m->eps_buf, B)) return rc;
        if (int rc = launch_axpy_out(x, m->eps_buf, sqrtf(1.0f - al[L - 1]), n, s)) return rc;
    }
    return m->ctx->f16x2 ? mcvd_ctx_check_range(m->ctx) : 0;      // f16x2 range guard: a non-finite epsilon anywhere in the loop is an error
    API_CATCH
}

// F-PNDM (FPNDM_sampler, models/__init__.py:38-99 + models/pndm.py): the whole loop on the device.  Steps run upwards 0, skip, ...,
// t_next = the previous step (-1 first), alpha table = flipped alphas indexed by t + 1, network label = t, Runge-Kutta for the first
// three steps (4 evaluations, midpoint label (t + t_next) / 2 as a float), 4th-order Adams-Bashforth afterwards.  The scalar
// coefficients are evaluated in fp32, operation by operation, as torch evaluates the reference's 0-dim tensor expressions.
int mcvd_fpndm_run(mcvd_model* m, float* x, const float* cond, int subsample_steps, int flags, int B) {
    API_TRY
    MCVD_REQUIRE(m && x && B > 0, "fpndm_run: bad arguments");
    MCVD_REQUIRE(m->finalized, "fpndm_run before mcvd_model_finalize");
    if (int rc = mcvd_ctx_clear_range(m->ctx)) return rc;
    const int T = m->d.num_classes;
    MCVD_REQUIRE(subsample_steps > 0 && subsample_steps <= T, "fpndm_run: subsample_steps=%d (the reference divides by it)", subsample_steps);
    const int64_t per = (int64_t)m->d.channels * m->d.num_frames * m->d.image_size * m->d.image_size;
    const int64_t n = per * B;
    if (int rc = m->prepare_B(B)) return rc;
    if (int rc = m->prepare_cond(cond, B)) return rc;
    struct CacheGuard { mcvd_model* m; ~CacheGuard() { m->cond_cache_valid = false; m->uniform_labels = 0; } } guard{m};
    m->uniform_labels = 1;
    hipStream_t s = m->ctx->stream;
    if (m->fp_B < B) {
        MCVD_HIP_CHECK(hipStreamSynchronize(s));
        if (m->fp_buf) MCVD_HIP_CHECK(hipFree(m->fp_buf));
        m->fp_buf = nullptr;
        MCVD_HIP_CHECK(hipMalloc((void**)&m->fp_buf, (size_t)9 * n * sizeof(float)));
        m->fp_B = B;
    }
    float* ring[4] = {m->fp_buf, m->fp_buf + n, m->fp_buf + 2 * n, m->fp_buf + 3 * n};      // eps history, oldest first after rotation
    float* e2 = m->fp_buf + 4 * n; float* e3 = m->fp_buf + 5 * n; float* e4 = m->fp_buf + 6 * n;
    float* xt = m->fp_buf + 7 * n; float* comb = m->fp_buf + 8 * n;
    const int clip = (flags & MCVD_FLAG_CLIP_BEFORE) ? 1 : 0;
    const int skip = T / subsample_steps;                                                 // :60
    auto alpha_old = [&](int idx) { return m->alphas[T - 1 - idx]; };                     // alphas.flip(0)  :57
    auto transfer = [&](float* out, const float* xx, float t_from, float t_to, const float* et) -> int {     // pndm.py:19-33
        const float at = alpha_old((int)t_from + 1), an = alpha_old((int)t_to + 1);      // t.long() truncates toward zero
        const float d = an - at;
        const float sa = sqrtf(at);
        const float c1 = 1.0f / (sa * (sa + sqrtf(an)));
        const float c2 = 1.0f / (sa * (sqrtf((1.0f - an) * at) + sqrtf((1.0f - at) * an)));
        return launch_pndm_transfer(out, xx, et, d, c1, c2, clip, n, s);
    };
    auto model_i = [&](const float* xx, int t, float* out) -> int {
        if (int rc = launch_fill_labels(m->labels, t, B, s)) return rc;
        return m->forward(xx, m->labels, cond, out, B);
    };
    auto model_f = [&](const float* xx, float t, float* out) -> int {
        if (int rc = launch_fill_labels_f(m->labels_f, t, B, s)) return rc;
        m->labels_f32 = 1;
        const int rc = m->forward(xx, m->labels_f, cond, out, B);
        m->labels_f32 = 0;
        return rc;
    };
    int n_ets = 0;
    int t_prev = -1;
    int steps_left = 1 << 30;
#ifdef MCVD_DIAG
    if (const char* ms_env = getenv("MCVD_FPNDM_MAXSTEPS")) steps_left = atoi(ms_env);          // diagnostics build: stop after this many steps
#endif
    for (int t = 0; t < T && steps_left > 0; t += skip, --steps_left) {
        MCVD_REQUIRE(t + 1 < T, "fpndm_run: index %d is out of bounds for the alpha table of size %d (the reference fails the same way "
                     "when num_classes is not a multiple of subsample_steps)", t + 1, T);
        const int t_next = t_prev;                                                        // steps_next = [-1] + steps[:-1]  :62
        const float t_mid = (float)(((double)t + (double)t_next) / 2.0);                  // pndm.py:42 (true division)
        if (n_ets > 2) {                                                                  // gen_order_4, pndm.py:44-47
            float* oldest = ring[0];                                                      // rotate: the oldest estimate is overwritten
            ring[0] = ring[1]; ring[1] = ring[2]; ring[2] = ring[3]; ring[3] = oldest;
            if (int rc = model_i(x, t, ring[3])) return rc;
            const float* in[4] = {ring[3], ring[2], ring[1], ring[0]};
            const float w[4] = {55.0f, -59.0f, 37.0f, -9.0f};
            if (int rc = launch_lincomb(comb, in, w, (float)(1.0 / 24.0), 4, n, s)) return rc;
        } else {                                                                          // runge_kutta, pndm.py:3-17
            float* e1 = ring[n_ets + 1 < 4 ? n_ets + 1 : 3];                              // slots 1,2,3 -> in age order once 3 are stored
            if (int rc = model_i(x, t, e1)) return rc;
            if (int rc = transfer(xt, x, (float)t, t_mid, e1)) return rc;
            if (int rc = model_f(xt, t_mid, e2)) return rc;
            if (int rc = transfer(xt, x, (float)t, t_mid, e2)) return rc;
            if (int rc = model_f(xt, t_mid, e3)) return rc;
            if (int rc = transfer(xt, x, (float)t, (float)t_next, e3)) return rc;
            if (int rc = model_i(xt, t_next, e4)) return rc;
            const float* in[4] = {e1, e2, e3, e4};
            const float w[4] = {1.0f, 2.0f, 2.0f, 1.0f};
            if (int rc = launch_lincomb(comb, in, w, (float)(1.0 / 6.0), 4, n, s)) return rc;
            ++n_ets;
        }
        if (int rc = transfer(x, x, (float)t, (float)t_next, comb)) return rc;            // pndm.py:51 (in place: elementwise)
        t_prev = t;
    }
    return m->ctx->f16x2 ? mcvd_ctx_check_range(m->ctx) : 0;      // f16x2 range guard
    API_CATCH
}

int mcvd_sampler_update(mcvd_ctx* ctx, int kind, float* x, const float* eps, const float* noise, float c_x0a, float c_x0b,
                        float c_mean0, float c_mean1, float c_noise, int clip, int64_t n) {
    MCVD_REQUIRE(ctx && x && eps, "sampler_update: NULL argument");
    return launch_sampler_update(kind, x, eps, noise, c_x0a, c_x0b, c_mean0, c_mean1, c_noise, clip, n, 0, 0, 0, 0, 4,
                                 ctx->stream);
}

int mcvd_pack_frames_u8(mcvd_ctx* ctx, const float* frames01, uint8_t* out, int B, int T, int C, int H, int W) {
    MCVD_REQUIRE(ctx, "pack_frames_u8: NULL ctx");
    return launch_pack_frames_u8(frames01, out, B, T, C, H * W, ctx->stream);
}

int mcvd_gamma_noise(mcvd_ctx* ctx, float* out, const float* raw, float k, float theta, float kt, float sd, uint64_t seed,
                     uint64_t sample_offset, uint64_t draw, int B, int64_t per_sample) {
    MCVD_REQUIRE(ctx && out, "gamma_noise: NULL argument");
    return launch_gamma_noise(out, raw, k, theta, kt, sd, seed, sample_offset, draw, B, per_sample, ctx->stream);
}

int mcvd_randn(mcvd_ctx* ctx, float* out, uint64_t seed, uint64_t sample_offset, uint64_t draw, int B, int64_t per_sample) {
    MCVD_REQUIRE(ctx && out, "randn: NULL argument");
    return launch_randn(out, seed, sample_offset, draw, B, per_sample, ctx->stream);
}

// ------------------------------------------------------------------------------------------------ stand-alone ops
int mcvd_upfirdn2d(mcvd_ctx* ctx, const float* in, const float* kernel_host, int kh, int kw, int up, int down, int pad0,
                   int pad1, float* out, int N, int C, int H, int W) {
    API_TRY
    MCVD_REQUIRE(ctx && in && kernel_host && out, "upfirdn2d: NULL argument");
    MCVD_REQUIRE(up >= 1 && down >= 1 && kh >= 1 && kw >= 1, "upfirdn2d: up/down/kernel");
    const int oh = (H * up + pad0 + pad1 - kh) / down + 1, ow = (W * up + pad0 + pad1 - kw) / down + 1;
    MCVD_REQUIRE(oh > 0 && ow > 0, "upfirdn2d: empty output");
    if (int rc = ctx->ensure_scratch((size_t)kh * kw * sizeof(float))) return rc;
    MCVD_HIP_CHECK(hipMemcpyAsync(ctx->scratch, kernel_host, (size_t)kh * kw * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    MCVD_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return launch_upfirdn2d(in, ctx->scratch, kh, kw, up, down, pad0, pad1, out, N * C, H, W, oh, ow, ctx->stream);
    API_CATCH
}

int mcvd_op_conv2d(mcvd_ctx* ctx, const float* x0, int C0, const float* x1, int C1, const float* w, const float* bias,
                   int Cout, int ks, const float* coef, int act, const float* res, float out_scale, float* y, int B, int H,
                   int W) {
    API_TRY
    MCVD_REQUIRE(ctx && x0 && w && bias && y, "op_conv2d: NULL argument");
    MCVD_REQUIRE(ks == 1 || ks == 3, "op_conv2d: ks=%d", ks);
    ConvArgs a{};
    a.x0 = x0; a.x1 = x1; a.C0 = C0; a.C1 = x1 ? C1 : 0;
    a.coef = coef; a.act = act; a.res = res; a.out_scale = out_scale; a.y = y;
    a.B = B; a.Cin = a.C0 + a.C1; a.Cout = Cout; a.H = H; a.W = W; a.ks = ks;
    a.cot = conv_cout_tile(Cout);
    a.CinP = round_up(a.Cin, conv_chunk(ks));
    a.CoutP = round_up(Cout, 32 * a.cot);
    const size_t wfloats = (size_t)a.CinP * ks * ks * a.CoutP;
    const bool wino = (ctx->conv_shape == 4 || ctx->conv_shape == 8 || (ctx->conv_shape >= 10 && ctx->conv_shape <= 13) ||
                       (ctx->conv_shape >= 16 && ctx->conv_shape <= 20)) && conv_wino_supported(ks, H, W);
    const bool wino_h = wino && (ctx->conv_shape == 12 || ctx->conv_shape == 13);     // fp16 pieces as well
    const bool wino_b = wino && (ctx->conv_shape == 10 || ctx->conv_shape == 11 || (ctx->conv_shape >= 16 && ctx->conv_shape <= 20));     // bf16 pieces as well
    const int np1 = (ks == 1 && ctx->conv_shape == 14) ? 2 : (ks == 1 && ctx->conv_shape == 15) ? 3 : 0;      // 1x1 pieces
    const size_t ufloats = wino ? (size_t)a.CinP * 16 * a.CoutP : 0;
    const size_t hfloats = wino_h ? (size_t)((conv_wino2h_weight_floats(a.CinP, a.CoutP) + 3) / 4 * 4)
                           : wino_b ? (size_t)((conv_wino3_weight_floats(a.CinP, a.CoutP) + 3) / 4 * 4)
                           : np1 ? (size_t)((conv1x1_h2_weight_floats(a.CinP, a.CoutP, np1) + 3) / 4 * 4) : 0;
    const int kparts = (ctx->conv_shape == 8 || ctx->conv_shape == 11 || ctx->conv_shape == 13 || ctx->conv_shape == 17) ? 2
                       : (ctx->conv_shape == 18 || ctx->conv_shape == 20) ? 4 : ctx->conv_shape == 19 ? 8 : 0;
    const size_t pfloats = wino ? (size_t)kparts * B * Cout * H * W : 0;     // K-split partial results
    if (int rc = ctx->ensure_scratch((wfloats + a.CoutP + ufloats + hfloats + pfloats) * sizeof(float))) return rc;
    MCVD_HIP_CHECK(hipMemsetAsync(ctx->scratch, 0, (wfloats + a.CoutP + ufloats + hfloats) * sizeof(float), ctx->stream));
    if (wino) {
        if (int rc = launch_pack_wino_weight(w, ctx->scratch + wfloats + a.CoutP, Cout, a.Cin, a.CinP, a.CoutP, ctx->stream)) return rc;
        a.wpw = ctx->scratch + wfloats + a.CoutP;
        if (wino_h) {
            if (int rc = launch_pack_wino2h_weight(w, ctx->scratch + wfloats + a.CoutP + ufloats, Cout, a.Cin, a.CinP, a.CoutP, ctx->stream)) return rc;
            a.wph = ctx->scratch + wfloats + a.CoutP + ufloats;
        }
        if (wino_b) {
            if (int rc = launch_pack_wino3_weight(w, ctx->scratch + wfloats + a.CoutP + ufloats, Cout, a.Cin, a.CinP, a.CoutP, ctx->stream)) return rc;
            a.wpb = ctx->scratch + wfloats + a.CoutP + ufloats;
        }
        if (pfloats) { a.part = ctx->scratch + wfloats + a.CoutP + ufloats + hfloats; a.part_floats = pfloats; }
    }
    if (int rc = launch_pack_conv_weight(w, ctx->scratch, Cout, a.Cin, ks, a.CinP, a.CoutP, 0, 0, ctx->stream)) return rc;
    if (np1) {
        if (int rc = launch_pack_conv1x1_h2(ctx->scratch, ctx->scratch + wfloats + a.CoutP, a.CinP, a.CoutP, ctx->stream, np1)) return rc;
        (np1 == 2 ? a.wph : a.wpb) = ctx->scratch + wfloats + a.CoutP;
    }
    MCVD_HIP_CHECK(hipMemcpyAsync(ctx->scratch + wfloats, bias, (size_t)Cout * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
    a.wp = ctx->scratch;
    a.bias = ctx->scratch + wfloats;
    a.shape_hint = ctx->conv_shape;
    if ((ctx->conv_shape == 5 || ctx->conv_shape == 6 || ctx->conv_shape == 9 || ctx->conv_shape == 14 || ctx->conv_shape == 15) && ctx->conv_cot > 0) a.cot = ctx->conv_cot;
    a.wdma = ctx->conv_wdma;
    a.pgrid = ctx->persist_grid;
    a.dbg = ctx->dbg;
    a.stats = ctx->naive_conv ? nullptr : ctx->stats_buf;
    if (ctx->spade_gb) {
        MCVD_REQUIRE(wino && !ctx->naive_conv && coef && act, "op_conv2d: the SPADE prologue exists in the Winograd kernel only (conv_shape 4 / 8, "
                     "3x3, coef and act given)");
        a.gb = ctx->spade_gb;
        a.coef2 = ctx->spade_coef2;
        ConvArgs t = a;
        t.ksplit = ctx->conv_shape == 8 ? 2 : 0;
        MCVD_REQUIRE(conv_wino_usable(t) || (t.ksplit = 0, conv_wino_usable(t)), "op_conv2d: shape not served by the Winograd kernel");
    }
    return ctx->naive_conv ? launch_conv_naive(a, ctx->stream) : launch_conv_mfma(a, ctx->stream);
    API_CATCH
}

int mcvd_last_conv_kernel(void) { return last_conv_kernel(); }

int mcvd_last_conv_stats_np(void) { return last_conv_stats_np(); }

int mcvd_op_gn_finalize(mcvd_ctx* ctx, const float* st0, int C0, int np0, const float* st1, int C1, int np1, int groups, float eps,
                        int mode, const float* p0, const float* p1, int emb_stride, int emb_off, float* coef_out, int B, int HW) {
    MCVD_REQUIRE(ctx && st0 && coef_out, "op_gn_finalize: NULL argument");
    GnArgs a{};
    a.C0 = C0; a.C1 = st1 ? C1 : 0; a.groups = groups; a.eps = eps; a.mode = mode; a.p0 = p0; a.p1 = p1;
    a.emb_stride = emb_stride; a.emb_off = emb_off; a.coef = coef_out; a.B = B; a.HW = HW;
    return launch_gn_finalize(a, st0, np0, st1, np1, ctx->stream);
}

int mcvd_model_graph_stats(mcvd_model* m, int64_t* captures, int64_t* replays) {
    MCVD_REQUIRE(m, "graph_stats: NULL model");
    if (captures) *captures = m->graph_captures;
    if (replays) *replays = m->graph_replays;
    return 0;
}

int mcvd_op_gn_coef(mcvd_ctx* ctx, const float* x0, int C0, const float* x1, int C1, int groups, float eps, int mode,
                    const float* p0, const float* p1, int emb_stride, int emb_off, float* coef_out, int B, int HW) {
    MCVD_REQUIRE(ctx && x0 && coef_out, "op_gn_coef: NULL argument");
    GnArgs a{};
    a.x0 = x0; a.x1 = x1; a.C0 = C0; a.C1 = x1 ? C1 : 0; a.groups = groups; a.eps = eps; a.mode = mode; a.p0 = p0; a.p1 = p1;
    a.emb_stride = emb_stride; a.emb_off = emb_off; a.coef = coef_out; a.B = B; a.HW = HW;
    return launch_gn_coef(a, ctx->stream);
}

int mcvd_op_attention(mcvd_ctx* ctx, const float* qkv, float* out, int B, int C, int heads, int HW) {
    MCVD_REQUIRE(ctx && qkv && out, "op_attention: NULL argument");
    const bool shared = ctx->share_fence && mcvd_ctx_shares_device(ctx);
    return launch_attention(ctx->naive_attn, shared ? 0 : ctx->f16x2, shared ? 0 : ctx->bf16x3, qkv, out, B, C, heads, HW, ctx->stream);
}

int mcvd_op_fir2(mcvd_ctx* ctx, const float* x, const float* coef, int act, int up, float* y, int B, int C, int H, int W) {
    MCVD_REQUIRE(ctx && x && y, "op_fir2: NULL argument");
    return launch_fir2(x, coef, act, up, y, B, C, H, W, nullptr, nullptr, nullptr, nullptr, ctx->stream, ctx->fir_form);
}

}  // extern "C"
