// Library-level entry points: version / error text and the hipGraph helpers used to replay a whole
// training step (a few dozen short kernels; at B=256 the step is launch-bound).
#include "common.h"

extern "C" int amid_version(void) { return 100; }

extern "C" const char* amid_error_string(int code) {
    if (code == AMID_OK) return "ok";
    if (code == AMID_ERR_ARG) return "amid: bad argument (null pointer or inconsistent size)";
    if (code == AMID_ERR_UNSUPPORTED) return "amid: shape not supported by the gfx950 kernels";
    return hipGetErrorString((hipError_t)code);
}

extern "C" int amid_device_sync(void) { return (int)hipDeviceSynchronize(); }

// ---- stream capture -> executable graph ---------------------------------------------------------
extern "C" int amid_graph_capture_begin(void* stream) {
    return (int)hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal);
}

extern "C" int amid_graph_capture_end(void* stream, void** graph_exec_out) {
    AMID_CHECK_ARG(graph_exec_out);
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture((hipStream_t)stream, &g);
    if (e != hipSuccess) return (int)e;
    hipGraphExec_t ex = nullptr;
    e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    const hipError_t e2 = hipGraphDestroy(g);
    if (e != hipSuccess) return (int)e;
    if (e2 != hipSuccess) return (int)e2;
    *graph_exec_out = (void*)ex;
    return AMID_OK;
}

extern "C" int amid_graph_launch(void* graph_exec, void* stream) {
    AMID_CHECK_ARG(graph_exec);
    return (int)hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream);
}

extern "C" int amid_graph_destroy(void* graph_exec) {
    if (!graph_exec) return AMID_OK;
    return (int)hipGraphExecDestroy((hipGraphExec_t)graph_exec);
}

// ---- HIP events on an arbitrary stream (bench.py times kernels on the stream they run on) --------
extern "C" int amid_event_create(void** ev_out) {
    AMID_CHECK_ARG(ev_out);
    hipEvent_t e;
    hipError_t r = hipEventCreate(&e);
    if (r != hipSuccess) return (int)r;
    *ev_out = (void*)e;
    return AMID_OK;
}
extern "C" int amid_event_record(void* ev, void* stream) { return (int)hipEventRecord((hipEvent_t)ev, (hipStream_t)stream); }
extern "C" int amid_event_sync(void* ev) { return (int)hipEventSynchronize((hipEvent_t)ev); }
extern "C" int amid_event_elapsed_ms(void* start, void* stop, float* ms_out) {
    AMID_CHECK_ARG(ms_out);
    return (int)hipEventElapsedTime(ms_out, (hipEvent_t)start, (hipEvent_t)stop);
}
extern "C" int amid_event_destroy(void* ev) { return (int)hipEventDestroy((hipEvent_t)ev); }
