// aps_internal.h — shared plumbing of libaps_hip.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/aps.h"

namespace aps {

// ---- error transport ---------------------------------------------------------------------------
struct Error : std::exception {
    int code;
    std::string msg;
    Error(int c, std::string m) : code(c), msg(std::move(m)) {}
    const char* what() const noexcept override { return msg.c_str(); }
};

[[noreturn]] void fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
void set_last_error(const char* s);

#define APS_HIP(expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            ::aps::fail(e_ == hipErrorOutOfMemory ? APS_E_OOM : APS_E_DEVICE,               \
                        "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__,    \
                        __LINE__);                                                          \
    } while (0)

#define APS_REQUIRE(cond, code, ...)                   \
    do {                                               \
        if (!(cond)) ::aps::fail((code), __VA_ARGS__); \
    } while (0)

// Wraps the body of every extern "C" entry point.
template <class F>
inline int guarded(F&& f) {
    try {
        f();
        return APS_OK;
    } catch (const Error& e) {
        set_last_error(e.msg.c_str());
        return e.code;
    } catch (const std::bad_alloc&) {
        set_last_error("host allocation failed");
        return APS_E_OOM;
    } catch (const std::exception& e) {
        set_last_error(e.what());
        return APS_E_INTERNAL;
    } catch (...) {
        set_last_error("unknown exception");
        return APS_E_INTERNAL;
    }
}

// ---- per-thread context --------------------------------------------------------------------------
struct Ctx {
    int device = -1;
    hipStream_t own_stream = nullptr;
    hipStream_t user_stream = nullptr;
    bool use_user_stream = false;
    int own_priority = 0;  // 1: own_stream was created at the device's highest priority
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    struct Block {
        void* p;
        size_t bytes;
        bool used;
    };
    std::vector<Block> blocks;
    hipStream_t stream() const { return use_user_stream ? user_stream : own_stream; }
    // auxiliary streams of this thread for fork/join sections (many small independent launch chains, e.g. the
    // per-image descriptor preparation of a pair batch), with one event each; created on first use
    std::vector<hipStream_t> aux;
    std::vector<hipEvent_t> aux_ev;
    hipEvent_t fork_ev = nullptr;
};

// Fork/join on the calling thread's auxiliary streams: fork(n) makes streams 0..n-1 wait for everything queued on stream()
// so far and returns them; join() makes stream() wait for everything queued on them since.
std::vector<hipStream_t>& aux_fork(int n);
void aux_join(int n);
// Scope of a fork/join section.  join() is the normal exit; when the scope is left by an exception instead, the
// auxiliary streams are drained on the host before the caller's workspace blocks (which their kernels still use) go
// back to the pool.
struct AuxScope {
    int n;
    bool joined = false;
    std::vector<hipStream_t>& streams;
    explicit AuxScope(int n_) : n(n_), streams(aux_fork(n_)) {}
    AuxScope(const AuxScope&) = delete;
    AuxScope& operator=(const AuxScope&) = delete;
    void join() {
        joined = true;
        aux_join(n);
    }
    ~AuxScope() {
        if (!joined)
            for (int i = 0; i < n && i < (int)streams.size(); ++i) (void)hipStreamSynchronize(streams[i]);
    }
};

// Makes sure a gfx950 device is selected for this thread and returns the context.
Ctx& ctx();
inline hipStream_t stream() { return ctx().stream(); }

// Cached device workspace (hipMalloc is synchronising; steady state must not call it).
void* ws_alloc(size_t bytes);
void ws_free(void* p);

template <class T>
struct Ws {
    T* p = nullptr;
    size_t n = 0;
    Ws() = default;
    explicit Ws(size_t count) { alloc(count); }
    Ws(const Ws&) = delete;
    Ws& operator=(const Ws&) = delete;
    Ws(Ws&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr; }
    Ws& operator=(Ws&& o) noexcept {
        if (this != &o) {
            reset();
            p = o.p;
            n = o.n;
            o.p = nullptr;
        }
        return *this;
    }
    ~Ws() { reset(); }
    void alloc(size_t count) {
        reset();
        n = count;
        p = static_cast<T*>(ws_alloc((count ? count : 1) * sizeof(T)));
    }
    void reset() {
        if (p) ws_free(p);
        p = nullptr;
    }
    T* get() const { return p; }
    operator T*() const { return p; }
};

// ---- host/device pointer staging -----------------------------------------------------------------
bool is_device_ptr(const void* p);

// Input that may live on the host: gives a device pointer valid until destruction.
template <class T>
struct In {
    const T* d = nullptr;
    Ws<T> tmp;
    In() = default;
    In(const T* p, size_t count) { bind(p, count); }
    void bind(const T* p, size_t count) {
        if (count == 0 || p == nullptr) {
            d = p;
            return;
        }
        if (is_device_ptr(p)) {
            d = p;
        } else {
            tmp.alloc(count);
            APS_HIP(hipMemcpyAsync(tmp.p, p, count * sizeof(T), hipMemcpyHostToDevice, stream()));
            d = tmp.p;
        }
    }
    const T* get() const { return d; }
    operator const T*() const { return d; }
};

// Output that may live on the host: device pointer now, copied back by commit().
template <class T>
struct Out {
    T* d = nullptr;
    T* host = nullptr;
    size_t n = 0;
    Ws<T> tmp;
    Out() = default;
    Out(T* p, size_t count) { bind(p, count); }
    void bind(T* p, size_t count) {
        n = count;
        if (p == nullptr) return;
        if (count == 0 || is_device_ptr(p)) {
            d = p;
        } else {
            host = p;
            tmp.alloc(count);
            d = tmp.p;
        }
    }
    // Copies back `count` elements (default: all).  Synchronises when the target is host memory.
    void commit(size_t count = SIZE_MAX) {
        if (!host) return;
        if (count > n) count = n;
        if (count) {
            APS_HIP(hipMemcpyAsync(host, d, count * sizeof(T), hipMemcpyDeviceToHost, stream()));
            APS_HIP(hipStreamSynchronize(stream()));
        }
    }
    T* get() const { return d; }
    operator T*() const { return d; }
    bool present() const { return d != nullptr; }
};

// Scoped event bracket around a kernel launch site (no-op unless aps_profile_enable(1)).
struct Prof {
    int slot = -1;
    void* ev_b = nullptr;
    explicit Prof(const char* name);
    ~Prof();
};

inline void check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) fail(APS_E_DEVICE, "launch of %s failed: %s", what, hipGetErrorString(e));
}

// match.hip: the screening half of the blocked global k-NN (see the definition)
int64_t screened_block_top3(const float* X_dev, int64_t ld, int layout, const std::vector<int64_t>& block_off,
                            std::vector<int64_t>& job_off, uint32_t* t3_idx, float* t3_d, float* t3_b);
// The pooled matcher's search with featureMatchingGlobal's filter in view (match.hip): rows the filter provably drops
// at `ratio` are flagged in dismissed[] and not searched; for the others t3_* hold, per (row, image) slot, the certified
// prefix of the three nearest rows of that image and the bound of its unlisted rows.
int64_t screened_global_top3(const float* X_dev, int64_t ld, int layout, const std::vector<int64_t>& img_off, float ratio,
                             std::vector<int64_t>& job_off, uint32_t* t3_idx, float* t3_d, float* t3_b, uint8_t* dismissed,
                             int64_t* n_survivors, int img_a = 0, int img_b = -1,  // [img_a, img_b): the query images of this call
                             struct GlobalPrep* keep = nullptr);  // the images' operand forms, built by the first pass and reused by the next
// (a search in several passes over ranges of query images prepares the column sets once: global_prep_new / _free own them)
struct GlobalPrep* global_prep_new();
void global_prep_free(struct GlobalPrep* p);

inline unsigned cdiv(size_t a, size_t b) { return static_cast<unsigned>((a + b - 1) / b); }

// compile-time loop: the body receives std::integral_constant<int, I>, so register arrays are indexed by constants
// by construction (an index the optimiser fails to fold sends the whole array to scratch memory)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}


}  // namespace aps
