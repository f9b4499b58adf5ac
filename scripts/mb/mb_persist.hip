// Stand-alone harness for the persistent recurrence kernels (csrc/persist.hip compiled into this translation unit):
//   * correctness: a short decode against a host (double precision) LSTM on the same random weights;
//   * timing: T steps, HIP-event time per step, and (PS_PROF builds) the in-kernel timeline of workgroup 0.
//   python scripts/mb/instrument.py apply   (csrc/persist.hip + instrumentation/persist.hip.patch -> gen/persist.hip: the stage clocks live in the patch)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -DPS_PROF=1024 -I ../../multilingual_text_to_speech_amd/csrc -I ../../include -o mb_persist mb_persist.hip && ./mb_persist [B] [T]
#include "gen/persist.hip"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdarg.h>
#include <vector>


thread_local char g_mtts_err[512] = {0};
int mtts_fail(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_mtts_err, sizeof(g_mtts_err), fmt, ap); va_end(ap); return 1; }

static std::vector<float> host_rand(size_t n, float scale) {
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = scale * ((float)rand() / RAND_MAX - 0.5f);
    return h;
}
static float* to_dev(const std::vector<float>& h) {
    float* d; (void)hipMalloc(&d, h.size() * 4); (void)hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    return d;
}
static float* dev_zero(size_t n) { float* d; (void)hipMalloc(&d, n * 4); (void)hipMemset(d, 0, n * 4); return d; }

// layout of mtts_lstm_pack_weights (fp32): [16-column group c][k-block][half][lane][4], unit-major gate columns
static std::vector<float> pack_lstm(const std::vector<float>& W, int H, int K) {
    const int nkb = K / 32;
    std::vector<float> out((size_t)4 * H * K);
    for (int c = 0; c < H / 4; ++c)
        for (int kb = 0; kb < nkb; ++kb)
            for (int h = 0; h < 2; ++h)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 4; ++e) {
                        const int i = lane & 15, q = lane >> 4, unit = 4 * c + (i >> 2), gate = i & 3;
                        out[((((size_t)c * nkb + kb) * 2 + h) * 64 + lane) * 4 + e] = W[((size_t)gate * H + unit) * K + 32 * kb + 8 * q + 4 * h + e];
                    }
    return out;
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 64, T = argc > 2 ? atoi(argv[2]) : 240, H = 1024, Dm = 544, L = 120, A = 128;
    const int Tc = 5;                                            // steps of the correctness run
    DecoderArgs a; memset(&a, 0, sizeof(a));
    a.B = B; a.L = L; a.T = T; a.H = H; a.Dm = Dm; a.A = A; a.fast = 1; a.training = 1; a.p_hidden = 0.1f;
    const std::vector<float> W = host_rand((size_t)4 * H * H, 0.12f), bias = host_rand(4 * H, 0.2f), pre = host_rand((size_t)T * B * 4 * H, 2.f);
    a.gen_w2p = to_dev(pack_lstm(W, H, H)); a.gen_bias_u = to_dev(bias); a.pre_gen = to_dev(pre);
    a.h_gen = dev_zero((size_t)(T + 1) * B * H); a.c_gen = dev_zero((size_t)(T + 1) * B * H); a.gates_gen = dev_zero((size_t)T * B * 4 * H);
    const long wsb = mtts_decoder_persist_ws_bytes(B, L, H, Dm, A);
    (void)hipMalloc(&a.persist_ws, wsb); (void)hipMemset(a.persist_ws, 0, wsb); a.persist_ws_bytes = wsb;
#ifdef PS_PROF
    unsigned long long* prof; (void)hipMalloc(&prof, PS_PROF * 8); (void)hipMemset(prof, 0, PS_PROF * 8);
    g_ps_prof = prof;
#endif
    // ---- correctness: Tc steps vs a host LSTM (bias / pre / columns unit-major: column 4 u + gate, gates i f g o)
    if (pgen_launch(a, 0, Tc, 0)) { printf("pgen_launch failed: %s\n", g_mtts_err); return 1; }
    std::vector<float> hd((size_t)(Tc + 1) * B * H), cd((size_t)(Tc + 1) * B * H);
    (void)hipMemcpy(hd.data(), a.h_gen, hd.size() * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(cd.data(), a.c_gen, cd.size() * 4, hipMemcpyDeviceToHost);
    printf("status after the correctness run: %d\n", mtts_decoder_persist_status(a.persist_ws, 0));
    {
        std::vector<double> h((size_t)B * H, 0.0), c((size_t)B * H, 0.0), hn((size_t)B * H);
        double worst = 0;
        for (int t = 0; t < Tc; ++t) {
            for (int b = 0; b < B; ++b)
                for (int u = 0; u < H; ++u) {
                    double g[4];
                    for (int q = 0; q < 4; ++q) {
                        double s = (double)bias[4 * u + q] + (double)pre[((size_t)t * B + b) * 4 * H + 4 * u + q];
                        const float* w = &W[((size_t)q * H + u) * H];
                        for (int k = 0; k < H; ++k) s += h[(size_t)b * H + k] * (double)w[k];
                        g[q] = s;
                    }
                    const double ig = 1 / (1 + exp(-g[0])), fg = 1 / (1 + exp(-g[1])), gg = tanh(g[2]), og = 1 / (1 + exp(-g[3]));
                    const double cn = fg * c[(size_t)b * H + u] + ig * gg;
                    c[(size_t)b * H + u] = cn; hn[(size_t)b * H + u] = og * tanh(cn);
                }
            h = hn;
            for (size_t i = 0; i < h.size(); ++i) {
                worst = fmax(worst, fabs(h[i] - (double)hd[(size_t)(t + 1) * B * H + i]));
                worst = fmax(worst, fabs(c[i] - (double)cd[(size_t)(t + 1) * B * H + i]));
            }
        }
        printf("pgen B=%d: max |h, c - host double| over %d steps = %.3e  %s\n", B, Tc, worst, worst < 2e-5 ? "OK" : "MISMATCH");
    }
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0, 0);
        if (pgen_launch(a, 0, T, 0)) { printf("pgen_launch failed: %s\n", g_mtts_err); return 1; }
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("pgen7 B=%d T=%d: %.3f ms = %.2f us per step (status %d)\n", B, T, ms, ms * 1e3 / T, mtts_decoder_persist_status(a.persist_ws, 0));
    }
#ifdef PS_PROF
    std::vector<unsigned long long> hp_gen(PS_PROF); (void)hipMemcpy(hp_gen.data(), prof, PS_PROF * 8, hipMemcpyDeviceToHost);
#endif
    // ---- pdec timing (random operands; parity is covered by the pytest suite)
    {
        const int P = 256, K = Dm + H;
        a.T = T; a.P = P; a.ksz = 31; a.C = 32;
        a.att_w2p = to_dev(host_rand((size_t)4 * H * K, 0.08f)); a.att_bias_u = to_dev(host_rand(4 * H, 0.2f)); a.att_w_pre_u = a.att_bias_u;
        a.pre_att = to_dev(host_rand((size_t)T * B * 4 * H, 2.f));
        a.h_att = dev_zero((size_t)(T + 1) * B * H); a.c_att = dev_zero((size_t)(T + 1) * B * H); a.gates_att = dev_zero((size_t)T * B * 4 * H);
        a.w_query = to_dev(host_rand((size_t)A * H, 0.1f)); a.memory = to_dev(host_rand((size_t)B * L * Dm, 1.f)); a.Mt = to_dev(host_rand((size_t)B * L * A, 1.f));
        a.U = to_dev(host_rand((size_t)A * 31, 0.3f)); a.att_bias = to_dev(host_rand(A, 0.1f)); a.w_energy = to_dev(host_rand(A, 0.5f));
        std::vector<int> lens(B, L); int* dl; (void)hipMalloc(&dl, B * 4); (void)hipMemcpy(dl, lens.data(), B * 4, hipMemcpyHostToDevice); a.lengths = dl;
        a.ctx = dev_zero((size_t)(T + 1) * B * Dm); a.cum = dev_zero((size_t)(T + 1) * B * L); a.align = dev_zero((size_t)T * B * L); a.q_all = dev_zero((size_t)T * B * A);
        // ---- every hand-off form must reproduce the barrier-only form bit for bit: sentinel-polled h row, early h-part loads, both
        struct Form { const char* name; bool poll, early; };
        const Form forms[] = {{"barrier        ", false, false}, {"polled h       ", true, false}, {"barrier + early", false, true}, {"polled + early ", true, true}};
        {
            const int Tq = T < 40 ? T : 40;
            struct Buf { const char* name; float* d; size_t n; };
            const Buf bufs[] = {{"h", a.h_att, (size_t)(Tq + 1) * B * H}, {"c", a.c_att, (size_t)(Tq + 1) * B * H}, {"gates", a.gates_att, (size_t)Tq * B * 4 * H},
                                {"ctx", a.ctx, (size_t)(Tq + 1) * B * Dm}, {"cum", a.cum, (size_t)(Tq + 1) * B * L}, {"align", a.align, (size_t)Tq * B * L},
                                {"q", a.q_all, (size_t)Tq * B * A}};
            std::vector<std::vector<float>> ref;
            for (int pass = 0; pass < 4; ++pass) {
                for (const Buf& b : bufs) (void)hipMemset(b.d, 0, b.n * 4);
                g_pdec_poll_off = !forms[pass].poll; g_pdec_early_off = !forms[pass].early;
                if (pdec_launch(a, 0, Tq, 0)) { printf("pdec_launch failed: %s\n", g_mtts_err); return 1; }
                (void)hipDeviceSynchronize();
                printf("pdec (%s) correctness run: status %d\n", forms[pass].name, mtts_decoder_persist_status(a.persist_ws, 0));
                size_t total_bad = 0;
                for (size_t i = 0; i < sizeof(bufs) / sizeof(bufs[0]); ++i) {
                    std::vector<float> h(bufs[i].n);
                    (void)hipMemcpy(h.data(), bufs[i].d, bufs[i].n * 4, hipMemcpyDeviceToHost);
                    if (pass == 0) {
                        double asum = 0; size_t nan = 0;
                        for (float v : h) { asum += fabs((double)v); nan += v != v; }
                        printf("  %-6s %9zu floats, mean |x| %.4f, %zu NaN\n", bufs[i].name, h.size(), asum / h.size(), nan);
                        ref.push_back(h); continue;
                    }
                    size_t bad = 0, first = 0;
                    for (size_t k = 0; k < h.size(); ++k)
                        if (memcmp(&h[k], &ref[i][k], 4) != 0 && bad++ == 0) first = k;
                    if (bad) printf("  %-6s %zu differ from the barrier form (first at %zu: %.9g vs %.9g)  MISMATCH\n", bufs[i].name, bad, first, h[first], ref[i][first]);
                    total_bad += bad;
                }
                if (pass) printf("  -> %zu values differ from the barrier form  %s\n", total_bad, total_bad ? "MISMATCH" : "OK");
            }
        }
        for (int variant = 0; variant < 4; ++variant) {
            g_pdec_poll_off = !forms[variant].poll; g_pdec_early_off = !forms[variant].early;
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipEventRecord(e0, 0);
                if (pdec_launch(a, 0, T, 0)) { printf("pdec_launch failed: %s\n", g_mtts_err); return 1; }
                (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                printf("pdec (%s) B=%d T=%d: %.3f ms = %.2f us per step (status %d)\n", forms[variant].name, B, T, ms, ms * 1e3 / T, mtts_decoder_persist_status(a.persist_ws, 0));
            }
        }
#ifdef PS_PROF
        std::vector<unsigned long long> hq(PS_PROF); (void)hipMemcpy(hq.data(), prof, PS_PROF * 8, hipMemcpyDeviceToHost);
        printf("pdec workgroup 0 / thread 0 (cycles): gates | cell+publish | barrier | h,q | energies | exchange | softmax | context | barrier || step\n");
        for (int st = 20; st < 28; ++st) {
            const unsigned long long* q = &hq[10 * st];
            printf("  step %3d:", st);
            for (int k = 0; k < 9; ++k) printf(" %6llu", q[k + 1] - q[k]);
            printf(" || %6llu\n", q[10] - q[0]);
        }
#endif
    }
#ifdef PS_PROF
    std::vector<unsigned long long>& hp = hp_gen;
    printf("workgroup 0 / wave 5, shader cycles per group-step (MFMA + partial sums | wait + issue | total):\n");
    for (int i = 40; i < 52; ++i)
        printf("  gs %3d: %6llu | %6llu | %6llu\n", i, hp[2 * i + 1] - hp[2 * i], hp[2 * i + 2] - hp[2 * i + 1], hp[2 * i + 2] - hp[2 * i]);
    printf("service wave of group 0 (workgroup 0): wait for partial sums | cell + publish | drain + arrive | state stores + land | step\n");
    for (int st = 20; st < 28; ++st) {
        const unsigned long long* q = &hp[512 + 4 * st];
        printf("  step %3d: %6llu | %6llu | %6llu | %6llu | %6llu\n", st, q[0] - q[-1], q[1] - q[0], q[2] - q[1], q[3] - q[2], q[4] - q[0]);
    }
#endif
    return 0;
}
