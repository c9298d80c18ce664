// runtime.hip -- error reporting, stream / graph / event helpers of the C ABI.
#include <stdarg.h>
#include <string.h>
#include "nd_common.h"

static thread_local char g_err[512] = "";

void nd_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

#define ND_HIP(call)                                                        \
    do {                                                                    \
        hipError_t e_ = (call);                                             \
        if (e_ != hipSuccess) {                                             \
            nd_set_error("%s: %s", #call, hipGetErrorString(e_));           \
            return (int)e_;                                                 \
        }                                                                   \
    } while (0)

int nd_current_device() {
    int dev = 0;
    return hipGetDevice(&dev) == hipSuccess ? dev : 0;
}

int nd_device_cus() {
    static std::atomic<int> cache[64];                       // zero-initialised; 0 = not queried yet
    const int dev = nd_current_device();
    int cus = (dev >= 0 && dev < 64) ? cache[dev].load(std::memory_order_relaxed) : 0;
    if (cus > 0) return cus;
    hipDeviceProp_t prop;
    cus = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    if (dev >= 0 && dev < 64) cache[dev].store(cus, std::memory_order_relaxed);
    return cus;
}

extern "C" int nd_version(void) { return 1000 * 0 + 2; }
extern "C" const char* nd_last_error(void) { return g_err; }

extern "C" int nd_device_arch(char* buf, int n) {
    ND_REQUIRE(buf && n > 0, ND_E_BADARG, "nd_device_arch: bad buffer");
    int dev = 0;
    ND_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    ND_HIP(hipGetDeviceProperties(&prop, dev));
    strncpy(buf, prop.gcnArchName, n - 1);
    buf[n - 1] = 0;
    return 0;
}

extern "C" int nd_stream_create(void** stream) {
    ND_REQUIRE(stream, ND_E_BADARG, "nd_stream_create: null");
    hipStream_t s;
    ND_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (void*)s;
    return 0;
}
extern "C" int nd_stream_destroy(void* stream) { ND_HIP(hipStreamDestroy((hipStream_t)stream)); return 0; }
extern "C" int nd_stream_sync(void* stream) { ND_HIP(hipStreamSynchronize((hipStream_t)stream)); return 0; }

// Graph and event calls act on a stream that may belong to another device than the thread's current one (one host thread driving the shards of several
// GPUs: GaussianDiffusion.sample under nn.DataParallel(device_ids=[...]), models/modules.py:73-83): they make the stream's device current for the call.
struct nd_stream_device_guard {
    int prev = -1;
    bool switched = false;
    explicit nd_stream_device_guard(void* stream) {
        int dev = -1;
        if (stream && hipStreamGetDevice((hipStream_t)stream, &dev) == hipSuccess && hipGetDevice(&prev) == hipSuccess && dev >= 0 && dev != prev)
            switched = hipSetDevice(dev) == hipSuccess;
        (void)hipGetLastError();
    }
    ~nd_stream_device_guard() { if (switched) (void)hipSetDevice(prev); }
};

extern "C" int nd_stream_device(void* stream) {
    int dev = -1;
    if (!stream || hipStreamGetDevice((hipStream_t)stream, &dev) != hipSuccess) { (void)hipGetLastError(); return -1; }
    return dev;
}

extern "C" int nd_graph_begin(void* stream) {
    nd_stream_device_guard g(stream);
    ND_HIP(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
    return 0;
}
extern "C" int nd_graph_end(void* stream, void** graph_exec) {
    ND_REQUIRE(graph_exec, ND_E_BADARG, "nd_graph_end: null");
    nd_stream_device_guard dg(stream);
    hipGraph_t g = nullptr;
    ND_HIP(hipStreamEndCapture((hipStream_t)stream, &g));
    hipGraphExec_t ge = nullptr;
    hipError_t e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) {
        nd_set_error("hipGraphInstantiate: %s", hipGetErrorString(e));
        return (int)e;
    }
    *graph_exec = (void*)ge;
    return 0;
}
extern "C" int nd_graph_launch(void* graph_exec, void* stream) {
    ND_REQUIRE(graph_exec, ND_E_STATE, "nd_graph_launch: null graph");
    nd_stream_device_guard dg(stream);
    ND_HIP(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream));
    return 0;
}
extern "C" int nd_graph_destroy(void* graph_exec) {
    if (graph_exec) ND_HIP(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
    return 0;
}

extern "C" int nd_event_create(void** ev) {
    ND_REQUIRE(ev, ND_E_BADARG, "nd_event_create: null");
    hipEvent_t e;
    ND_HIP(hipEventCreate(&e));
    *ev = (void*)e;
    return 0;
}
extern "C" int nd_event_record(void* ev, void* stream) { ND_HIP(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream)); return 0; }
// `stream` waits for `ev` (recorded on another stream): the fork / join edges of the two-branch step graph
extern "C" int nd_stream_wait_event(void* stream, void* ev) { ND_HIP(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)ev, 0)); return 0; }
// an event without timing: a pure dependency (cheaper to record, and the only kind a captured cross-stream edge needs)
extern "C" int nd_event_create_untimed(void** ev) {
    ND_REQUIRE(ev, ND_E_BADARG, "nd_event_create_untimed: null");
    hipEvent_t e;
    ND_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    *ev = (void*)e;
    return 0;
}
extern "C" int nd_event_elapsed_ms(void* start, void* stop, float* ms) {
    ND_REQUIRE(ms, ND_E_BADARG, "nd_event_elapsed_ms: null");
    ND_HIP(hipEventSynchronize((hipEvent_t)stop));
    ND_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return 0;
}
extern "C" int nd_event_destroy(void* ev) { ND_HIP(hipEventDestroy((hipEvent_t)ev)); return 0; }
