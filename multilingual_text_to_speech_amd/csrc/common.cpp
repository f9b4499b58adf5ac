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
        default: return -1;
    }
}
