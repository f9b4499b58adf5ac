// Error plumbing + version for libmtts_hip.
#include "common.h"
#include <stdarg.h>

thread_local char g_mtts_err[512] = {0};

int mtts_fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_mtts_err, sizeof(g_mtts_err), fmt, ap);
    va_end(ap);
    return 1;
}

MTTS_API const char* mtts_last_error(void) { return g_mtts_err; }
MTTS_API int mtts_version(void) { return 100; }

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
        default: return -1;
    }
}

// ---- side stream + event pool for the two-chain decoder schedules (created once; streams/events are not memory) ----
static hipStream_t g_side = nullptr;
static hipEvent_t g_events[256];
static int g_event_next = 0, g_event_count = 0;

hipStream_t side_stream() {
    if (!g_side) {
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) lo = 0;          // lo = least priority
        if (hipStreamCreateWithPriority(&g_side, hipStreamNonBlocking, lo) != hipSuccess) g_side = nullptr;
    }
    return g_side;
}

// second helper stream (least priority): weight-gradient GEMMs of finished chunks, off both decoder chains
static hipStream_t g_wgrad = nullptr;
hipStream_t wgrad_stream() {
    if (!g_wgrad) {
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) lo = 0;
        if (hipStreamCreateWithPriority(&g_wgrad, hipStreamNonBlocking, lo) != hipSuccess) g_wgrad = nullptr;
    }
    return g_wgrad;
}

hipEvent_t pool_event() {
    if (g_event_count < 256) {
        if (hipEventCreateWithFlags(&g_events[g_event_count], hipEventDisableTiming) == hipSuccess) return g_events[g_event_count++];
        return nullptr;      // hipEventRecord(nullptr) reports the failure to the caller
    }
    hipEvent_t e = g_events[g_event_next];
    g_event_next = (g_event_next + 1) % 256;
    return e;
}
