// Stand-alone harness of the attention-step backward (csrc/attention_bwd.hip compiled into this translation unit): per-launch time over
// back-to-back launches on random operands at the benchmark's shape, and (-DATB_PROF) the in-kernel stage timeline of workgroup 0.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -DATB_PROF -o mb_attn_bwd mb_attn_bwd.hip && ./mb_attn_bwd [B] [L] [Dm] [n_part]
#include "gen/attention_bwd.hip"
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
    const int B = argc > 1 ? atoi(argv[1]) : 64, L = argc > 2 ? atoi(argv[2]) : 120, Dm = argc > 3 ? atoi(argv[3]) : 544, NP = argc > 4 ? atoi(argv[4]) : 7;
    const int A = 128, ksz = 31, nch = 4, NREP = 200;
    AttnBwdArgs p; memset(&p, 0, sizeof(p));
    p.q = dev_rand((size_t)B * A, 1.f); p.Mt = dev_rand((size_t)B * L * A, 1.f); p.U = dev_rand((size_t)A * ksz, 0.3f);
    p.bias = dev_rand(A, 0.1f); p.v = dev_rand(A, 0.5f); p.memory = dev_rand((size_t)B * L * Dm, 1.f); p.ctx = dev_rand((size_t)B * Dm, 1.f);
    std::vector<int> lens(B, L); int* dl; (void)hipMalloc(&dl, B * 4); (void)hipMemcpy(dl, lens.data(), B * 4, hipMemcpyHostToDevice); p.lengths = dl;
    p.w = dev_rand((size_t)B * L, 0.02f); p.cum_in = dev_rand((size_t)B * L, 0.3f); p.dcum_out = dev_rand((size_t)B * L, 1.f);
    p.dcum_in = dev_rand((size_t)B * L, 0.f); p.dctx = dev_rand((size_t)B * Dm, 1.f); p.dctx_total = dev_rand((size_t)B * Dm, 0.f);
    p.part = dev_rand((size_t)NP * B * Dm, 1.f); p.n_part = NP; p.part_ks = (long)B * Dm; p.part_ld = Dm;
    p.dq = dev_rand((size_t)B * A, 0.f); p.dMt = dev_rand((size_t)B * L * A, 0.f); p.dU_slab = dev_rand((size_t)B * nch * A * ksz, 0.f);
    p.dv_slab = dev_rand((size_t)B * nch * A, 0.f); p.dbias_slab = dev_rand((size_t)B * nch * A, 0.f);
    p.B = B; p.L = L; p.A = A; p.Dm = Dm; p.ksz = ksz; p.nch = nch;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0, 0);
        for (int i = 0; i < NREP; ++i) if (mtts_attn_step_bwd(&p, 0)) { printf("launch failed: %s\n", g_mtts_err); return 1; }
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("attn_bwd B=%d L=%d Dm=%d n_part=%d: %.2f us per launch (%d workgroups)\n", B, L, Dm, NP, ms * 1e3 / NREP, B * nch);
    }
#ifdef ATB_PROF
    unsigned long long st[16];
    (void)hipMemcpyFromSymbol(st, HIP_SYMBOL(g_atb_stamps), sizeof(st));
    const char* names[] = {"loads + LDS staging", "block sum S", "dw / de", "PL recompute + tanh + ds + dMt", "dq / slabs / dU contraction", "g contraction + window", "dcum atomics"};
    printf("workgroup 0 / thread 0, shader cycles per stage (last launch):");
    for (int k = 0; k < 7; ++k) printf(" %s %llu |", names[k], st[k + 1] - st[k]);
    printf(" total %llu\n", st[7] - st[0]);
#endif
    return 0;
}
