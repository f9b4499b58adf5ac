// Stand-alone timing harness for the fused LSTM step kernels of large batches (csrc/lstm_step.hip compiled into this translation unit):
// per-launch time over back-to-back launches, with compile-time knock-outs that isolate the streams of lstm_fused_kernel (F) and
// lstm_fused2_kernel (F2):
//   -DLF_NO_X     activation fragments not loaded (constant operand)      -DLF_NO_W     weight fragments not loaded
//   -DLF_NO_MFMA  products skipped                                        -DLF_NO_EPI   no cell / query epilogue (accumulators stored raw)
//   -DLF_NO_STAGE (F2) no staging of the next block (no wait for its loads, no split, no LDS stores)
//   python scripts/mb/instrument.py apply   (csrc/lstm_step.hip + instrumentation/lstm_step.hip.patch -> gen/lstm_step.hip: the knock-outs live in the patch)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast [-DLF_...] -I ../../multilingual_text_to_speech_amd/csrc -I ../../include -o mb_lstm_fused mb_lstm_fused.hip
//   ./mb_lstm_fused [B] [Kctx] [prec: 0 fp32 MFMA, 1 bf16, 2 fp32 as pre-split bf16 planes] [nb_max: 0 = F2 for prec 1 / 2 (lone chain), 4 = F]
#include "gen/lstm_step.hip"
#include <cstdio>
#include <cstdlib>
#include <stdarg.h>
#include <vector>

thread_local char g_mtts_err[512] = {0};
int mtts_fail(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_mtts_err, sizeof(g_mtts_err), fmt, ap); va_end(ap); return 1; }

static float* dev_rand(size_t n, float scale) {
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = scale * ((float)rand() / (float)RAND_MAX - 0.5f);
    float* d; (void)hipMalloc(&d, n * 4); (void)hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    return d;
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 240, Kc = argc > 2 ? atoi(argv[2]) : 288, prec = argc > 3 ? atoi(argv[3]) : 0, nbmax = argc > 4 ? atoi(argv[4]) : 0;
    const int H = 1024, A = 128, K = Kc + H, NREP = 200;
    LstmStepArgs a; memset(&a, 0, sizeof(a));
    float* w = dev_rand((size_t)4 * H * K, 0.1f);
    void* wp; (void)hipMalloc(&wp, mtts_lstm_packed_weight_bytes(H, K, prec));
    float* b_ih = dev_rand(4 * H, 0.1f); float* bias_u; (void)hipMalloc(&bias_u, 4 * H * 4);
    LstmPackArgs pk; memset(&pk, 0, sizeof(pk));
    pk.w[0] = w; pk.K[0] = K; pk.ldw[0] = K; pk.nseg = 1; pk.H = H; pk.precision = prec; pk.dst = wp; pk.b_ih = b_ih; pk.b_hh = b_ih; pk.bias_u = bias_u;
    if (mtts_lstm_pack_weights(&pk, 0)) { printf("pack failed: %s\n", g_mtts_err); return 1; }
    int n = 0;
    if (Kc > 0) { a.x[n] = dev_rand((size_t)B * Kc, 1.f); a.K[n] = Kc; a.ldx[n] = Kc; ++n; }
    a.x[n] = dev_rand((size_t)B * H, 1.f); a.K[n] = H; a.ldx[n] = H; ++n;
    a.nseg = n; a.w_packed = wp; a.precision = prec; a.B = B; a.H = H; a.nb_max = nbmax;
    a.partials = dev_rand((size_t)mtts_lstm_step_partial_floats(B, H, K), 0.f);
    a.pre = dev_rand((size_t)B * 4 * H, 1.f); a.ldpre = 4 * H; a.bias_u = bias_u;
    a.c_prev = dev_rand((size_t)B * H, 1.f);
    a.h_out = dev_rand((size_t)B * H, 0.f); a.c_out = dev_rand((size_t)B * H, 0.f); a.gates_out = dev_rand((size_t)B * 4 * H, 0.f);
    a.w_query = dev_rand((size_t)A * H, 0.1f); a.A = A; a.qpart = dev_rand((size_t)(H / 16) * B * A, 0.f);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0, 0);
        for (int i = 0; i < NREP; ++i) if (lstm_step_launch(a, 0)) { printf("launch failed: %s\n", g_mtts_err); return 1; }
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("lstm step B=%d K=%d+%d prec=%d: %.2f us per launch\n", B, Kc, H, prec, ms * 1e3 / NREP);
    }
    return 0;
}
