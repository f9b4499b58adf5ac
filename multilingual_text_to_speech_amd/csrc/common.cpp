// Error plumbing + version for libmtts_hip.
#include "common.h"
#include <stdarg.h>
#include <stdlib.h>

thread_local char g_mtts_err[512] = {0};

int mtts_fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_mtts_err, sizeof(g_mtts_err), fmt, ap);
    va_end(ap);
    return 1;
}

MTTS_API const char* mtts_last_error(void) { return g_mtts_err; }
MTTS_API int mtts_version(void) { return 102; }      // 102: DecoderGradArgs.part_ring (see mtts.h)

// Bitmask of compile-time switches that make a build produce WRONG results by design (timing experiments).  The product sources have
// none (round 4 removed the last two); the entry stays so that bindings keep refusing a library that reports anything but 0
// (multilingual_text_to_speech_amd._C.lib() does unless MTTS_ALLOW_DEBUG_LIB=1).
MTTS_API int mtts_build_flags(void) { return 0; }

// sizeof() of the ABI structs, in header order, so that bindings can verify their mirrors.
MTTS_API int mtts_sizeof_struct(int which) {
    switch (which) {
        case 0: return (int)sizeof(GemmArgs);
        case 1: return (int)sizeof(BnArgs);
        case 2: return (int)sizeof(SkSeg);
        case 3: return (int)sizeof(SkinnyArgs);
        case 4: return (int)sizeof(AttnStepArgs);
        case 5: return (int)sizeof(DecoderArgs);
        case 6: return (int)sizeof(BiLstmArgs);
        case 7: return (int)sizeof(AttnBwdArgs);
        case 8: return (int)sizeof(DecoderGradArgs);
        case 9: return (int)sizeof(BiLstmGradArgs);
        case 10: return (int)sizeof(TacoLossArgs);
        case 11: return (int)sizeof(AdamArgs);
        case 12: return (int)sizeof(LstmPackArgs);
        case 13: return (int)sizeof(LstmStepArgs);
        case 14: return (int)sizeof(GenParamsArgs);
        default: return -1;
    }
}

// ---- per-(device, caller stream) library state ----------------------------------------------------------------------
// The two-chain decoder schedules need helper streams, ordering events and a split-K scratch arena.  All of it hangs off
// the (device, stream) the caller passed in, so that two models / threads driving different streams (or devices) of one
// process never share a helper stream, an event or scratch memory.  Streams and events are created on first use and
// kept for the life of the process (they are not memory); the scratch arena is always provided by the caller.
#include <mutex>
#include <vector>

namespace {
struct GraphEntry {            // one captured decoder range: exact argument block + its executable graph
    DecoderArgs args;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    int seen = 0;              // capture on the second sighting (the first run executes eagerly: one-time setup stays out of graphs)
};
struct StreamCtx {
    std::vector<GraphEntry*> graphs;
    int device = 0;
    hipStream_t owner = nullptr;
    hipStream_t side = nullptr, wgrad = nullptr;
    std::vector<hipEvent_t> events;      // ring, grown when its oldest event has not completed yet
    size_t next = 0;
    float* ws = nullptr;                 // per-stream override of the device default (mtts_set_stream_workspace)
    size_t ws_bytes = 0;
};
struct DeviceWs { int device; float* ptr; size_t bytes; };
std::mutex g_mu;
std::vector<StreamCtx*> g_ctx;
std::vector<DeviceWs> g_dev_ws;

int current_device() { int d = 0; if (hipGetDevice(&d) != hipSuccess) d = 0; return d; }

// caller must hold g_mu.  A helper stream resolves to the context of the stream it was created for.
StreamCtx* ctx_locked(hipStream_t s) {
    const int dev = current_device();
    for (StreamCtx* c : g_ctx)
        if (c->device == dev && (c->owner == s || (s && (c->side == s || c->wgrad == s)))) return c;
    StreamCtx* c = new StreamCtx();
    c->device = dev; c->owner = s;
    g_ctx.push_back(c);
    return c;
}

// Helper streams get the least priority (higher ones were measured with no effect on the train step, round 2).
hipStream_t low_priority_stream() {
    int lo = 0, hi = 0;
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) lo = hi = 0;      // lo = least priority (numerically largest)
    hipStream_t st = nullptr;
    if (hipStreamCreateWithPriority(&st, hipStreamNonBlocking, lo) != hipSuccess) st = nullptr;
    return st;
}
}  // namespace

// low-priority helper stream of caller stream `s`: generator-LSTM chain / reverse BiLSTM direction
hipStream_t side_stream(hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_mu);
    StreamCtx* c = ctx_locked(s);
    if (!c->side) c->side = low_priority_stream();
    return c->side;
}

// second helper stream (least priority): weight-gradient GEMMs of finished chunks, off both decoder chains
hipStream_t wgrad_stream(hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_mu);
    StreamCtx* c = ctx_locked(s);
    if (!c->wgrad) c->wgrad = low_priority_stream();
    return c->wgrad;
}

// Steps per chunk of the two-chain decoder schedules (chain hand-off, weight-gradient accumulation granularity).
// MTTS_CHUNK overrides the default of 48 (read per call: tests run the small fixtures with tiny chunks).
int decoder_chunk() {
    const char* e = getenv("MTTS_CHUNK");
    if (e && e[0]) { const int v = atoi(e); if (v >= 1 && v <= 4096) return v; }
    return 48;
}

// An ordering event from the ring of caller stream `s`.  An event is only handed out again once it has completed
// (hipEventQuery); otherwise the ring grows, so a long decode with tiny chunks can never re-record a pending event.
hipEvent_t pool_event(hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_mu);
    StreamCtx* c = ctx_locked(s);
    if (c->events.size() >= 64) {
        hipEvent_t e = c->events[c->next];
        if (hipEventQuery(e) == hipSuccess) { c->next = (c->next + 1) % c->events.size(); return e; }
        (void)hipGetLastError();      // hipErrorNotReady is sticky in hipGetLastError
    }
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;   // hipEventRecord(nullptr) reports it
    c->events.insert(c->events.begin() + c->next, e);
    c->next = (c->next + 1) % c->events.size();
    return e;
}

// Split-K scratch arena visible to launches on `s` (a helper stream sees its owner's arena).
float* workspace_for(hipStream_t s, size_t* bytes) {
    std::lock_guard<std::mutex> lk(g_mu);
    StreamCtx* c = ctx_locked(s);
    if (c->ws) { *bytes = c->ws_bytes; return c->ws; }
    for (const DeviceWs& d : g_dev_ws)
        if (d.device == c->device) { *bytes = d.bytes; return d.ptr; }
    *bytes = 0;
    return nullptr;
}

// Pack buffer of the pre-split GEMM core (gemm_planes.h) for launches on `s`: one per (device, stream) - helper streams get their own,
// their GEMMs run beside the caller's.  Either provided by the caller (mtts_set_planes_workspace: never resized, a request that does
// not fit returns nullptr and the GEMM takes the core that needs no pack pass) or owned by the library (hipMalloc, grow-only, 64 MB
// granules, released by mtts_planes_trim).  g_mu is NOT held across the stream synchronisation / hipFree / hipMalloc of a growth.
namespace {
struct PlanesBuf { int device; hipStream_t stream; char* ptr; size_t bytes; bool owned; };
std::vector<PlanesBuf> g_planes;
PlanesBuf* planes_entry_locked(int dev, hipStream_t s) {
    for (PlanesBuf& b : g_planes) if (b.device == dev && b.stream == s) return &b;
    g_planes.push_back(PlanesBuf{dev, s, nullptr, 0, true});
    return &g_planes.back();
}
}
char* planes_buffer(hipStream_t s, size_t bytes) {
    const int dev = current_device();
    char* old = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        PlanesBuf* e = planes_entry_locked(dev, s);
        if (e->bytes >= bytes) return e->ptr;
        if (!e->owned) return nullptr;               // the caller's arena is too small for this operand pair: no pack pass
        old = e->ptr; e->ptr = nullptr; e->bytes = 0;
    }
    if (old) {                                       // kernels of earlier launches on `s` may still read the old buffer
        if (hipStreamSynchronize(s) != hipSuccess) (void)hipGetLastError();
        (void)hipFree(old);
    }
    const size_t want = ((bytes + bytes / 4) + ((size_t)64 << 20) - 1) & ~(((size_t)64 << 20) - 1);
    void* q = nullptr;
    if (hipMalloc(&q, want) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    std::lock_guard<std::mutex> lk(g_mu);
    PlanesBuf* e = planes_entry_locked(dev, s);
    if (e->ptr || !e->owned) { (void)hipFree(q); return e->bytes >= bytes ? e->ptr : nullptr; }      // set meanwhile by another thread
    e->ptr = (char*)q; e->bytes = want;
    return e->ptr;
}

// Caller-provided pack arena for GEMMs launched on `stream` (nullptr / 0 hands the stream back to the library-owned buffer).
MTTS_API int mtts_set_planes_workspace(void* stream, void* ptr, size_t bytes) {
    char* old = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        PlanesBuf* e = planes_entry_locked(current_device(), (hipStream_t)stream);
        if (e->owned) old = e->ptr;
        e->ptr = (char*)ptr; e->bytes = ptr ? bytes : 0; e->owned = ptr == nullptr;
    }
    if (old) {
        if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) (void)hipGetLastError();
        (void)hipFree(old);
    }
    return 0;
}

// Release every library-owned pack buffer of the current device (their streams are synchronised first).  Returns the bytes freed.
MTTS_API size_t mtts_planes_trim(void) {
    std::vector<PlanesBuf> take;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        const int dev = current_device();
        for (PlanesBuf& b : g_planes)
            if (b.device == dev && b.owned && b.ptr) { take.push_back(b); b.ptr = nullptr; b.bytes = 0; }
    }
    size_t freed = 0;
    for (const PlanesBuf& b : take) {
        if (hipStreamSynchronize(b.stream) != hipSuccess) (void)hipGetLastError();      // a destroyed stream reports an error: nothing can be in flight on it
        if (hipFree(b.ptr) == hipSuccess) freed += b.bytes; else (void)hipGetLastError();
    }
    return freed;
}

MTTS_API int mtts_set_workspace(void* ptr, size_t bytes) {
    std::lock_guard<std::mutex> lk(g_mu);
    const int dev = current_device();
    for (DeviceWs& d : g_dev_ws)
        if (d.device == dev) { d.ptr = (float*)ptr; d.bytes = bytes; return 0; }
    g_dev_ws.push_back(DeviceWs{dev, (float*)ptr, bytes});
    return 0;
}

MTTS_API int mtts_set_stream_workspace(void* stream, void* ptr, size_t bytes) {
    std::lock_guard<std::mutex> lk(g_mu);
    StreamCtx* c = ctx_locked((hipStream_t)stream);
    c->ws = (float*)ptr; c->ws_bytes = bytes;
    return 0;
}

// ---- hipGraph replay of a free-running decoder range (BASELINE configs[4]: "hipGraph-captured decode steps") ----------------------
// mtts_decoder_fwd_graphed(args, stream): the launches of mtts_decoder_fwd(args) for a range [t0, t1) of the GENERAL schedule
// are captured once per distinct argument block (every pointer, size and the step range take part in the comparison) and replayed
// with ONE hipGraphLaunch afterwards.  The caller keeps the buffers alive and at the same addresses between calls
// (decoder_ops.GraphedDecode does); `stream` must not be the legacy default stream.  The first call with a new argument block
// runs eagerly, the second one captures, later ones replay.  Returns 0 on success; *replayed (nullable) says which path ran.
// Destroy every captured graph of `stream` (a caller that retires a set of fixed-address buffers calls this: the graphs keyed by
// those addresses can never be replayed again).  Returns the number of graphs destroyed.
MTTS_API int mtts_decoder_graphs_clear(void* stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    StreamCtx* c = ctx_locked((hipStream_t)stream);
    const int n = (int)c->graphs.size();
    for (GraphEntry* g : c->graphs) { if (g->exec) (void)hipGraphExecDestroy(g->exec); if (g->graph) (void)hipGraphDestroy(g->graph); delete g; }
    c->graphs.clear();
    return n;
}

MTTS_API int mtts_decoder_fwd_graphed(const DecoderArgs* args, void* stream, int* replayed) {
    hipStream_t s = (hipStream_t)stream;
    if (replayed) *replayed = 0;
    MTTS_REQUIRE(s != nullptr, "mtts_decoder_fwd_graphed: needs a non-default stream (stream capture)");
    MTTS_REQUIRE(!args->fast && !args->teacher && !args->frames_in, "mtts_decoder_fwd_graphed: free-running (general schedule) ranges only");
    GraphEntry* e = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        StreamCtx* c = ctx_locked(s);
        for (GraphEntry* g : c->graphs)
            if (memcmp(&g->args, args, sizeof(DecoderArgs)) == 0) { e = g; break; }
        if (!e) {
            if (c->graphs.size() >= 4096) {       // a long-lived process with ever-changing buffers: start over
                for (GraphEntry* g : c->graphs) { if (g->exec) (void)hipGraphExecDestroy(g->exec); if (g->graph) (void)hipGraphDestroy(g->graph); delete g; }
                c->graphs.clear();
            }
            e = new GraphEntry();
            memcpy(&e->args, args, sizeof(DecoderArgs));
            c->graphs.push_back(e);
        }
    }
    if (e->exec) {
        MTTS_CHECK_HIP(hipGraphLaunch(e->exec, s));
        if (replayed) *replayed = 1;
        return 0;
    }
    bool first;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        first = e->seen++ == 0;
    }
    if (first) return mtts_decoder_fwd(args, stream);
    MTTS_CHECK_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    const int rc = mtts_decoder_fwd(args, stream);
    hipGraph_t graph = nullptr;
    const hipError_t end = hipStreamEndCapture(s, &graph);
    if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    if (end != hipSuccess || !graph) return mtts_fail("hipStreamEndCapture: %s", hipGetErrorString(end));
    hipGraphExec_t exec = nullptr;
    const hipError_t inst = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    if (inst != hipSuccess) { (void)hipGraphDestroy(graph); return mtts_fail("hipGraphInstantiate: %s", hipGetErrorString(inst)); }
    e->graph = graph; e->exec = exec;
    MTTS_CHECK_HIP(hipGraphLaunch(exec, s));        // the captured launches have not executed yet
    if (replayed) *replayed = 1;
    return 0;
}
