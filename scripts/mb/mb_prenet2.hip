// Stand-alone harness of prenet2_kernel (csrc/lstm_step.hip compiled into this translation unit): the two prenet layers of one
// free-running frame, per-launch time over back-to-back launches and (-DPN_PROF) the stage timeline of workgroup (0, 0).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -DPN_PROF -o mb_prenet2 mb_prenet2.hip && ./mb_prenet2 [B] [M] [P]
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
    const int B = argc > 1 ? atoi(argv[1]) : 128, M = argc > 2 ? atoi(argv[2]) : 80, P = argc > 3 ? atoi(argv[3]) : 256;
    const int Mo = (M + 1 + 3) & ~3, NREP = 300;
    float* x = dev_rand((size_t)B * Mo, 1.f);
    float* w1 = dev_rand((size_t)P * M, 0.2f); float* b1 = dev_rand(P, 0.1f);       // (any values: the tile order only matters for the results)
    float* w2 = dev_rand((size_t)P * P, 0.1f); float* b2 = dev_rand(P, 0.1f);
    std::vector<uint8_t> hm((size_t)B * P);
    for (auto& v : hm) v = (rand() & 1);
    uint8_t *m1, *m2; (void)hipMalloc(&m1, hm.size()); (void)hipMalloc(&m2, hm.size());
    (void)hipMemcpy(m1, hm.data(), hm.size(), hipMemcpyHostToDevice); (void)hipMemcpy(m2, hm.data(), hm.size(), hipMemcpyHostToDevice);
    float* y1 = dev_rand((size_t)B * P, 0.f); float* y2 = dev_rand((size_t)B * P, 0.f);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0, 0);
        for (int i = 0; i < NREP; ++i)
            if (prenet2_launch(x, Mo, M, w1, b1, w2, b2, m1, m2, 2.f, y1, y2, B, P, 0) != 0) { printf("launch failed: %s\n", g_mtts_err); return 1; }
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("prenet2 B=%d M=%d P=%d: %.2f us per launch\n", B, M, P, ms * 1e3 / NREP);
    }
#ifdef PN_PROF
    unsigned long long st[8];
    (void)hipMemcpyFromSymbol(st, HIP_SYMBOL(g_pn_stamps), sizeof(st));
    printf("workgroup (0, 0) / thread 0, shader cycles: issue of the loads %llu | layer 1 (operands landed, MFMA, epilogue, LDS, barrier) %llu | layer-2 products %llu | "
           "K-half exchange + epilogue %llu | total %llu\n", st[1] - st[0], st[2] - st[1], st[3] - st[2], st[4] - st[3], st[4] - st[0]);
#endif
    return 0;
}
