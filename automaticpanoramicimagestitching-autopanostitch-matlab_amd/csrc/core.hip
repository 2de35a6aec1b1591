// core.hip — device selection, per-thread stream/workspace, error transport, library entry points.
#include <atomic>
#include <cstdlib>
#include <mutex>

#include <cstring>

#include "aps_internal.h"

namespace aps {

static thread_local std::string g_err;
static thread_local Ctx g_ctx;
// The device the most recent aps_set_device() call selected, in any thread.  A thread that has not chosen a device
// itself (worker pools of the host layer, parfor 'Threads' workers) starts on it instead of falling back to device 0.
static std::atomic<int> g_default_device{-1};

void set_last_error(const char* s) { g_err = s ? s : ""; }

void fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    throw Error(code, buf);
}

static bool device_is_gfx950(int dev) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return false;
    return std::strncmp(prop.gcnArchName, "gfx950", 6) == 0;
}

static int count_devices() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    int ok = 0;
    for (int i = 0; i < n; ++i)
        if (device_is_gfx950(i)) ++ok;
    return ok == n ? n : 0;  // mixed boxes are not supported: this library is gfx950-only
}

static void bind_device(Ctx& c, int dev) {
    int n = count_devices();
    if (n <= 0)
        fail(APS_E_DEVICE,
             "no gfx950 (MI355X) device is visible; libaps_hip has no CPU fallback by design");
    if (dev < 0 || dev >= n) fail(APS_E_ARG, "device %d out of range [0,%d)", dev, n);
    APS_HIP(hipSetDevice(dev));
    if (c.device != dev) {
        // drop everything tied to the previous device
        for (auto& b : c.blocks) (void)hipFree(b.p);
        c.blocks.clear();
        if (c.own_stream) (void)hipStreamDestroy(c.own_stream);
        if (c.ev0) (void)hipEventDestroy(c.ev0);
        if (c.ev1) (void)hipEventDestroy(c.ev1);
        // the auxiliary streams of aux_fork() and their events belong to the old device as well
        for (auto st : c.aux) (void)hipStreamDestroy(st);
        for (auto ev : c.aux_ev) (void)hipEventDestroy(ev);
        if (c.fork_ev) (void)hipEventDestroy(c.fork_ev);
        c.aux.clear();
        c.aux_ev.clear();
        c.fork_ev = nullptr;
        c.own_stream = nullptr;
        c.ev0 = c.ev1 = nullptr;
        c.device = dev;
    }
    if (!c.own_stream) {
        // APS_STREAM_PRIORITY=1: this thread's stream at the highest priority the device offers (an experiment switch: on
        // the 64 x 4K step it costs the resident extraction 2-3 ms and helps only where other streams' copies share the
        // runtime's hardware queues with the library's streams - GPU_MAX_HW_QUEUES=8, set by the Python package, is the
        // cure for that).
        int least = 0, greatest = 0;
        const char* e = std::getenv("APS_STREAM_PRIORITY");
        const bool want_high = e && std::atoi(e) != 0;
        if (want_high && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least) {
            APS_HIP(hipStreamCreateWithPriority(&c.own_stream, hipStreamNonBlocking, greatest));
            c.own_priority = 1;
        } else
            APS_HIP(hipStreamCreateWithFlags(&c.own_stream, hipStreamNonBlocking));
    }
    if (!c.ev0) APS_HIP(hipEventCreate(&c.ev0));
    if (!c.ev1) APS_HIP(hipEventCreate(&c.ev1));
}

Ctx& ctx() {
    Ctx& c = g_ctx;
    if (c.device < 0) {
        int dev = g_default_device.load(std::memory_order_acquire);
        if (dev < 0) {
            dev = 0;
            if (const char* e = std::getenv("APS_DEVICE")) dev = std::atoi(e);
        }
        bind_device(c, dev);
    } else {
        APS_HIP(hipSetDevice(c.device));
    }
    return c;
}

std::vector<hipStream_t>& aux_fork(int n) {
    Ctx& c = ctx();
    while ((int)c.aux.size() < n) {
        hipStream_t st = nullptr;
        hipEvent_t ev = nullptr;
        APS_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        APS_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        c.aux.push_back(st);
        c.aux_ev.push_back(ev);
    }
    if (!c.fork_ev) APS_HIP(hipEventCreateWithFlags(&c.fork_ev, hipEventDisableTiming));
    APS_HIP(hipEventRecord(c.fork_ev, c.stream()));
    for (int i = 0; i < n; ++i) APS_HIP(hipStreamWaitEvent(c.aux[i], c.fork_ev, 0));
    return c.aux;
}

void aux_join(int n) {
    Ctx& c = ctx();
    for (int i = 0; i < n && i < (int)c.aux.size(); ++i) {
        APS_HIP(hipEventRecord(c.aux_ev[i], c.aux[i]));
        APS_HIP(hipStreamWaitEvent(c.stream(), c.aux_ev[i], 0));
    }
}

void* ws_alloc(size_t bytes) {
    Ctx& c = ctx();
    bytes = (bytes + 255) & ~size_t(255);
    int best = -1;
    for (int i = 0; i < (int)c.blocks.size(); ++i) {
        auto& b = c.blocks[i];
        if (!b.used && b.bytes >= bytes && (best < 0 || b.bytes < c.blocks[best].bytes)) best = i;
    }
    if (best >= 0 && c.blocks[best].bytes <= 2 * bytes + (1u << 20)) {
        c.blocks[best].used = true;
        return c.blocks[best].p;
    }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
        // free cached blocks and retry once
        (void)hipGetLastError();
        for (auto it = c.blocks.begin(); it != c.blocks.end();) {
            if (!it->used) {
                (void)hipFree(it->p);
                it = c.blocks.erase(it);
            } else {
                ++it;
            }
        }
        e = hipMalloc(&p, bytes);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            fail(APS_E_OOM, "device allocation of %zu bytes failed: %s", bytes,
                 hipGetErrorString(e));
        }
    }
    c.blocks.push_back({p, bytes, true});
    return p;
}

void ws_free(void* p) {
    for (auto& b : g_ctx.blocks)
        if (b.p == p) {
            b.used = false;
            return;
        }
}

struct ProfRec {
    const char* name;
    hipEvent_t a, b;
};
// The profile store is process-wide (worker threads with their own streams all record into it); each record
// carries the two events of one launch site on the stream it ran on.
static std::atomic<int> g_prof_on{0};  // 0 off, 1 every launch site, 2 only the low-frequency sites (see aps.h)
static std::mutex g_prof_mu;
static std::vector<ProfRec> g_prof;
static std::vector<hipEvent_t> g_prof_pool;

static hipEvent_t prof_event_locked() {
    if (!g_prof_pool.empty()) {
        hipEvent_t e = g_prof_pool.back();
        g_prof_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    APS_HIP(hipEventCreate(&e));
    return e;
}

Prof::Prof(const char* name) {
    const int mode = g_prof_on.load(std::memory_order_relaxed);
    if (!mode) return;
    if (mode == 2) {
        // selective: the per-image / per-tile chains issue thousands of launches per stitch, and two event records
        // per launch are a measurable share of it (2.7 % of the 64-view step); the per-batch kernels are kept
        static const char* const keep[] = {"match", "ransac", "cover", "render_", "ba_", "crop", "gain", "knn", "global_"};
        bool ok = false;
        for (const char* k : keep) ok = ok || std::strncmp(name, k, std::strlen(k)) == 0;
        if (!ok) return;
    }
    hipStream_t st = stream();
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfRec r{name, prof_event_locked(), prof_event_locked()};
    APS_HIP(hipEventRecord(r.a, st));
    g_prof.push_back(r);
    ev_b = r.b;
    slot = 1;
}
Prof::~Prof() {
    if (slot >= 0) (void)hipEventRecord(static_cast<hipEvent_t>(ev_b), stream());
}

bool is_device_ptr(const void* p) {
    if (!p) return false;
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return false;  // plain pageable host memory
    }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

}  // namespace aps

using namespace aps;

extern "C" {

int aps_version(void) { return APS_VERSION; }

const char* aps_last_error(void) { return g_err.c_str(); }

int aps_device_count(void) { return count_devices(); }

int aps_set_device(int device) {
    return guarded([&] {
        bind_device(g_ctx, device);
        g_default_device.store(device, std::memory_order_release);
    });
}

int aps_set_thread_device(int device) {
    return guarded([&] { bind_device(g_ctx, device); });
}

int aps_set_thread_stream_priority(int level) {
    return guarded([&] {
        Ctx& c = ctx();
        int least = 0, greatest = 0;
        APS_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
        const int prio = level > 0 ? greatest : (level < 0 ? least : (least + greatest) / 2);
        hipStream_t fresh = nullptr;  // (the new stream first: a failure leaves the thread with the stream it had)
        APS_HIP(hipStreamCreateWithPriority(&fresh, hipStreamNonBlocking, prio));
        if (c.own_stream) {
            (void)hipStreamSynchronize(c.own_stream);
            (void)hipStreamDestroy(c.own_stream);
        }
        c.own_stream = fresh;
        c.own_priority = level > 0 ? 1 : 0;
    });
}

int aps_get_device(void) {
    int dev = -1;
    const int rc = guarded([&] { dev = ctx().device; });
    return rc == APS_OK ? dev : -1;
}

int aps_set_stream(void* hip_stream) {
    return guarded([&] {
        Ctx& c = ctx();
        c.user_stream = static_cast<hipStream_t>(hip_stream);
        c.use_user_stream = true;
        if (hip_stream == nullptr) c.use_user_stream = false;
    });
}

int aps_synchronize(void) {
    return guarded([&] { APS_HIP(hipStreamSynchronize(stream())); });
}

int aps_release_workspace(void) {
    return guarded([&] {
        Ctx& c = ctx();
        APS_HIP(hipStreamSynchronize(c.stream()));
        for (auto it = c.blocks.begin(); it != c.blocks.end();) {
            if (!it->used) {
                (void)hipFree(it->p);
                it = c.blocks.erase(it);
            } else {
                ++it;
            }
        }
    });
}

int aps_profile_enable(int on) {
    return guarded([&] {
        ctx();
        g_prof_on.store(on == 2 ? 2 : (on != 0 ? 1 : 0));
    });
}

int aps_profile_reset(void) {
    return guarded([&] {
        ctx();
        APS_HIP(hipDeviceSynchronize());
        std::lock_guard<std::mutex> lk(g_prof_mu);
        for (auto& r : g_prof) {
            g_prof_pool.push_back(r.a);
            g_prof_pool.push_back(r.b);
        }
        g_prof.clear();
    });
}

int aps_profile_get(const char* name, double* total_ms, int* launches) {
    return guarded([&] {
        APS_REQUIRE(name && total_ms && launches, APS_E_ARG, "NULL argument");
        ctx();
        APS_HIP(hipDeviceSynchronize());
        std::lock_guard<std::mutex> lk(g_prof_mu);
        double t = 0;
        int n = 0;
        for (auto& r : g_prof)
            if (std::strcmp(r.name, name) == 0) {
                float ms = 0;
                APS_HIP(hipEventElapsedTime(&ms, r.a, r.b));
                t += ms;
                ++n;
            }
        *total_ms = t;
        *launches = n;
    });
}

int aps_profile_series(const char* name, double* ms, int cap, int* count) {
    return guarded([&] {
        APS_REQUIRE(name && count && cap >= 0 && (cap == 0 || ms), APS_E_ARG, "bad argument");
        ctx();
        APS_HIP(hipDeviceSynchronize());
        std::lock_guard<std::mutex> lk(g_prof_mu);
        int n = 0;
        for (auto& r : g_prof)
            if (std::strcmp(r.name, name) == 0) {
                if (n < cap) {
                    float t = 0;
                    APS_HIP(hipEventElapsedTime(&t, r.a, r.b));
                    ms[n] = t;
                }
                ++n;
            }
        *count = n;
    });
}

int aps_profile_names(char* buf, int buf_len) {
    return guarded([&] {
        APS_REQUIRE(buf && buf_len > 0, APS_E_ARG, "bad buffer");
        std::string out;
        std::vector<const char*> seen;
        std::lock_guard<std::mutex> lk(g_prof_mu);
        for (auto& r : g_prof) {
            bool dup = false;
            for (auto* s : seen) dup |= std::strcmp(s, r.name) == 0;
            if (dup) continue;
            seen.push_back(r.name);
            if (!out.empty()) out += ';';
            out += r.name;
        }
        std::snprintf(buf, buf_len, "%s", out.c_str());
    });
}

int aps_timer_begin(void) {
    return guarded([&] {
        Ctx& c = ctx();
        APS_HIP(hipEventRecord(c.ev0, c.stream()));
    });
}

int aps_timer_end(float* ms) {
    return guarded([&] {
        APS_REQUIRE(ms != nullptr, APS_E_ARG, "ms is NULL");
        Ctx& c = ctx();
        APS_HIP(hipEventRecord(c.ev1, c.stream()));
        APS_HIP(hipEventSynchronize(c.ev1));
        APS_HIP(hipEventElapsedTime(ms, c.ev0, c.ev1));
    });
}

}  // extern "C"
