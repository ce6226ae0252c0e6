// Replay of a fixed call sequence as one device-side graph (include/aos2.h: aos2_capture_begin / _end, aos2_graph_launch).
// The per-frame body of Tracking::Track on its usual path -- Frame::Frame, TrackWithMotionModel, TrackLocalMap
// (src/Tracking.cc:268-520, 860-1039) -- is the same ~20 dependent launches for every frame of a sequence; for ONE sequence
// (batch 1) the time between them is a sixth of the frame's latency.  The calls are made once between aos2_capture_begin
// and aos2_capture_end on the stream of the Frame batch (nothing runs: HIP records the launches, and the other handles'
// streams join through the device-side waits the calls already make); aos2_graph_launch then enqueues the whole sequence
// with one call and the device runs the kernels back to back.
#include "aos2_common.h"

namespace aos2 {

// Every stream of the library is created here.  (Measured and dropped, round 5: compute units set aside for LocalBundleAdjustment's
// reduced-system kernel with hipExtStreamCreateWithCUMask, every other stream masked off them -- the kernel's workgroup needs a
// whole CU and waits for one while wide tracking kernels keep placing workgroups -- made the composite step 6 x longer, 7.5 ->
// 46 ms with 32 or 64 of the 256 CUs set aside: streams with a CU mask do not run beside each other the way plain ones do.)
int stream_create(hipStream_t *q, bool high_priority)
{
    int least = 0, greatest = 0;
    if (high_priority && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least)
        AOS2_HIP_CHECK(hipStreamCreateWithPriority(q, hipStreamNonBlocking, greatest));
    else
        AOS2_HIP_CHECK(hipStreamCreateWithFlags(q, hipStreamNonBlocking));
    return AOS2_OK;
}

}  // namespace aos2

struct aos2_graph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    size_t nodes = 0;
};

extern "C" int aos2_capture_begin(void *hip_stream)
{
    using namespace aos2;
    if (!hip_stream) {
        set_error("aos2_capture_begin: a stream of a handle is required (not the null stream)");
        return AOS2_ERR_ARG;
    }
    // relaxed: the calls in between may use the runtime freely (attribute queries, event creation) -- what they must not do
    // is wait on the host or allocate, which the handles only do on their FIRST call with a shape (run the sequence once before)
    AOS2_HIP_CHECK(hipStreamBeginCapture((hipStream_t)hip_stream, hipStreamCaptureModeRelaxed));
    return AOS2_OK;
}

extern "C" int aos2_capture_end(void *hip_stream, aos2_graph_t **out)
{
    using namespace aos2;
    if (!hip_stream || !out) {
        set_error("bad argument");
        return AOS2_ERR_ARG;
    }
    *out = nullptr;
    hipGraph_t g = nullptr;
    const hipError_t e = hipStreamEndCapture((hipStream_t)hip_stream, &g);
    if (e != hipSuccess || !g) {
        set_error("hipStreamEndCapture failed: %s (a call of the sequence waited on the host, allocated, or left another stream "
                  "unjoined: run the sequence once before capturing, end it with a call that orders `hip_stream` behind the rest)",
                  hipGetErrorString(e));
        (void)hipGetLastError();
        return AOS2_ERR_HIP;
    }
    aos2_graph *r = new aos2_graph();
    r->graph = g;
    (void)hipGraphGetNodes(g, nullptr, &r->nodes);
    const hipError_t ei = hipGraphInstantiate(&r->exec, g, nullptr, nullptr, 0);
    if (ei != hipSuccess) {
        set_error("hipGraphInstantiate failed: %s", hipGetErrorString(ei));
        (void)hipGraphDestroy(g);
        delete r;
        return AOS2_ERR_HIP;
    }
    *out = r;
    return AOS2_OK;
}

extern "C" int aos2_graph_launch(aos2_graph_t *g, void *hip_stream)
{
    using namespace aos2;
    if (!g || !g->exec) {
        set_error("bad argument");
        return AOS2_ERR_ARG;
    }
    AOS2_HIP_CHECK(hipGraphLaunch(g->exec, (hipStream_t)hip_stream));
    return AOS2_OK;
}

extern "C" int aos2_graph_nodes(const aos2_graph_t *g) { return g ? (int)g->nodes : 0; }

extern "C" void aos2_graph_destroy(aos2_graph_t *g)
{
    if (!g) return;
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
}
