// Runtime plumbing of libtabmat_hip.so: error strings, per-device workspace, thin
// memory / stream / event wrappers so a plain-C host (ctypes, cgo, JNI ...) can drive
// the kernels without any other HIP binding.
#include <stdarg.h>
#include <string.h>

#include <map>
#include <string>
#include <mutex>
#include <utility>

#include "common.hpp"

namespace tmh {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char *what, const char *file, int line) {
    set_error("HIP error %d (%s) at %s:%d in `%s`", (int)e, hipGetErrorString(e), file, line, what);
    return (int)e > 0 ? (int)e : TM_EINVAL;
}

// Tuning knobs (tm_tune_set): small integer settings that pick between launch geometries of the
// same kernel.  Defaults are the measured optima; the knobs exist so that a profile run can
// compare geometries inside one process.
static std::map<std::string, int64_t> g_tune;
static std::mutex g_tune_mu;

int64_t tune(const char *key, int64_t dflt) {
    std::lock_guard<std::mutex> lk(g_tune_mu);
    auto it = g_tune.find(key);
    return it == g_tune.end() ? dflt : it->second;
}

struct Workspace {
    void *ptr = nullptr;
    size_t bytes = 0;
    bool external = false;
};
static Workspace g_ws_ext[64];                              // tm_set_workspace, per device
static std::map<std::pair<int, hipStream_t>, Workspace> g_ws;   // library-owned, per (device, stream)
static std::mutex g_ws_mu;
static int64_t g_ws_generation = 0;   // bumped whenever a workspace pointer changes

int get_workspace(size_t bytes, void **ptr, hipStream_t st) {
    int dev = 0;
    TM_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) {
        set_error("device id %d out of range", dev);
        return TM_EINVAL;
    }
    std::lock_guard<std::mutex> lk(g_ws_mu);
    if (g_ws_ext[dev].external) {
        if (bytes > g_ws_ext[dev].bytes) {
            set_error("caller-provided workspace too small: need %zu bytes, have %zu", bytes,
                      g_ws_ext[dev].bytes);
            return TM_ENOMEM;
        }
        *ptr = g_ws_ext[dev].ptr;
        return TM_OK;
    }
    Workspace &w = g_ws[std::make_pair(dev, st)];
    if (bytes > w.bytes) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
            set_error("workspace of %zu bytes needed while the stream is being captured "
                      "(have %zu): run the op once before the capture", bytes, w.bytes);
            return TM_ENOMEM;
        }
        if (w.ptr) {
            TM_HIP(hipStreamSynchronize(st));
            TM_HIP(hipFree(w.ptr));
            w.ptr = nullptr;
            w.bytes = 0;
        }
        size_t want = bytes < (size_t(64) << 20) ? (size_t(64) << 20) : bytes + bytes / 4;
        TM_HIP(hipMalloc(&w.ptr, want));
        w.bytes = want;
        ++g_ws_generation;
    }
    *ptr = w.ptr;
    return TM_OK;
}

static bool g_prof_on = false;
static hipEvent_t g_prof_a = nullptr, g_prof_b = nullptr;
static bool g_prof_valid = false;
static bool g_prof_hold = false;     // prof_hold(true): the next prof_begin / prof_end pairs are ignored

void prof_hold(bool on) { g_prof_hold = on; }

void prof_begin(hipStream_t st) {
    if (!g_prof_on || g_prof_hold) return;
    if (!g_prof_a) {
        (void)hipEventCreate(&g_prof_a);
        (void)hipEventCreate(&g_prof_b);
    }
    (void)hipEventRecord(g_prof_a, st);
}

void prof_end(hipStream_t st) {
    if (!g_prof_on || g_prof_hold) return;
    (void)hipEventRecord(g_prof_b, st);
    g_prof_valid = true;
}

}  // namespace tmh

using namespace tmh;

extern "C" {

int tm_version(void) { return 100; }

const char *tm_last_error(void) { return g_err; }

int tm_device_count(int *count) {
    TM_REQUIRE(count != nullptr, "count is NULL");
    TM_HIP(hipGetDeviceCount(count));
    return TM_OK;
}

int tm_set_device(int device) {
    TM_HIP(hipSetDevice(device));
    return TM_OK;
}

int tm_device_info(char *name, int name_len, int *compute_units, int64_t *hbm_bytes) {
    int dev = 0;
    TM_HIP(hipGetDevice(&dev));
    hipDeviceProp_t p;
    TM_HIP(hipGetDeviceProperties(&p, dev));
    if (name && name_len > 0) {
        snprintf(name, (size_t)name_len, "%s (%s)", p.name, p.gcnArchName);
    }
    if (compute_units) *compute_units = p.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)p.totalGlobalMem;
    return TM_OK;
}

int tm_malloc(void **ptr, size_t bytes) {
    TM_REQUIRE(ptr != nullptr, "ptr is NULL");
    TM_HIP(hipMalloc(ptr, bytes ? bytes : 1));
    return TM_OK;
}

int tm_free(void *ptr) {
    if (ptr) TM_HIP(hipFree(ptr));
    return TM_OK;
}

int tm_memcpy_h2d(void *dst, const void *h_src, size_t bytes, void *stream) {
    TM_HIP(hipMemcpyAsync(dst, h_src, bytes, hipMemcpyHostToDevice, as_stream(stream)));
    return TM_OK;
}

int tm_memcpy_d2h(void *h_dst, const void *src, size_t bytes, void *stream) {
    TM_HIP(hipMemcpyAsync(h_dst, src, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
    TM_HIP(hipStreamSynchronize(as_stream(stream)));
    return TM_OK;
}

int tm_memset(void *dst, int value, size_t bytes, void *stream) {
    TM_HIP(hipMemsetAsync(dst, value, bytes, as_stream(stream)));
    return TM_OK;
}

int tm_stream_synchronize(void *stream) {
    TM_HIP(hipStreamSynchronize(as_stream(stream)));
    return TM_OK;
}

int tm_set_workspace(void *ptr, size_t bytes) {
    int dev = 0;
    TM_HIP(hipGetDevice(&dev));
    TM_REQUIRE(dev >= 0 && dev < 64, "device id out of range");
    std::lock_guard<std::mutex> lk(g_ws_mu);
    // library-owned workspaces of this device are released (their streams drained first)
    for (auto it = g_ws.begin(); it != g_ws.end();) {
        if (it->first.first == dev) {
            if (it->second.ptr) {
                TM_HIP(hipStreamSynchronize(it->first.second));
                TM_HIP(hipFree(it->second.ptr));
            }
            it = g_ws.erase(it);
        } else {
            ++it;
        }
    }
    ++g_ws_generation;
    Workspace &w = g_ws_ext[dev];
    w.ptr = ptr;
    w.bytes = ptr ? bytes : 0;
    w.external = ptr != nullptr;
    return TM_OK;
}

int tm_workspace_generation(int64_t *generation) {
    TM_REQUIRE(generation != nullptr, "generation is NULL");
    std::lock_guard<std::mutex> lk(g_ws_mu);
    *generation = g_ws_generation;
    return TM_OK;
}

int tm_tune_set(const char *h_key, int64_t value) {
    TM_REQUIRE(h_key != nullptr, "key is NULL");
    std::lock_guard<std::mutex> lk(g_tune_mu);
    if (value == INT64_MIN) g_tune.erase(h_key);     // back to the built-in default
    else g_tune[h_key] = value;
    return TM_OK;
}

int tm_tune_get(const char *h_key, int64_t dflt, int64_t *value) {
    TM_REQUIRE(h_key != nullptr && value != nullptr, "NULL argument");
    *value = tune(h_key, dflt);
    return TM_OK;
}

int tm_profile_enable(int on) {
    g_prof_on = on != 0;
    g_prof_valid = false;
    return TM_OK;
}

int tm_profile_last_ms(float *ms) {
    TM_REQUIRE(ms != nullptr, "ms is NULL");
    if (!g_prof_valid) {
        set_error("tm_profile_last_ms: no kernel recorded (call tm_profile_enable(1) first)");
        return TM_EINVAL;
    }
    TM_HIP(hipEventSynchronize(g_prof_b));
    TM_HIP(hipEventElapsedTime(ms, g_prof_a, g_prof_b));
    return TM_OK;
}

int tm_event_create(void **event) {
    TM_REQUIRE(event != nullptr, "event is NULL");
    hipEvent_t e;
    TM_HIP(hipEventCreate(&e));
    *event = (void *)e;
    return TM_OK;
}

int tm_event_destroy(void *event) {
    if (event) TM_HIP(hipEventDestroy((hipEvent_t)event));
    return TM_OK;
}

int tm_event_record(void *event, void *stream) {
    TM_HIP(hipEventRecord((hipEvent_t)event, as_stream(stream)));
    return TM_OK;
}

int tm_event_elapsed_ms(void *start, void *stop, float *ms) {
    TM_REQUIRE(ms != nullptr, "ms is NULL");
    TM_HIP(hipEventSynchronize((hipEvent_t)stop));
    TM_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return TM_OK;
}

}  // extern "C"
