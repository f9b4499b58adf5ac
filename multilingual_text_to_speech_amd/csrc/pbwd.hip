// Persistent decoder BACKWARD, chain A (round 6): the attention / attention-LSTM recurrence of a chunk of teacher-forced steps in ONE launch.
//
// Replaces, for 33..64 rows, chain A's three dependent launches per step of csrc/decoder_bwd.hip (reference: autograd through
// modules/tacotron2.py:180-198, modules/attention.py:39-86, modules/layers.py:37-47): {attention backward || dG_att W_hh^T} -> dq W_q +
// attention-LSTM cell backward -> dG_att W_ih[:, P:]^T.  Measured on the bench shape (profiles/r05_train_step_phases.txt): 33 us per step
// alone, 58 us beside chain B and the helper streams' GEMMs - every launch waits for a slot next to the 100-300 us GEMM workgroups and
// for the dispatcher.  Chain B (generator LSTM) and the per-chunk GEMMs keep their streams.
//
// Here 256 workgroups stay RESIDENT for a whole chunk (<= 128 VGPRs, <= 64 KiB LDS: one of them fits on a CU beside one GEMM
// workgroup of the helper streams) and walk the steps with three grid barriers per step:
//   stage 1  attention backward of step t (workgroup (b, ch), the body of attention_bwd_body.h)                          | barrier
//   stage 2  dq W_q + attention-LSTM cell backward of step t -> dG_att(t)                                                 | barrier
//   stage 3  dG_att(t) W_ih[:, P:]^T (ctx columns, K-split slabs: what the next attention backward waits for)             | arrive
//            ... dG_att(t) W_hh^T (h columns: only the NEXT cell backward reads them) in the shadow of that barrier       | wait
// The arithmetic is the per-step kernels' (the same bodies, the same K splits, the same summation order).
//
// Visibility between workgroups inside the launch (MI355X_MICROARCH.md): everything a stage hands to another workgroup is either an
// L2 atomic (dq, dcum: as in the per-step kernels) or a write-through (sc1) store into memory that no workgroup has read before in this
// launch - the partial slabs live in a RING indexed by the step (DecoderGradArgs.part_ring, >= chunk + 2 slots), dG's packed copy is
// per step anyway - so consumers use plain loads of lines that cannot be resident in their L1 / L2; every wave drains its stores
// (s_waitcnt vmcnt(0)) in front of a barrier's arrive.  State a thread hands to ITSELF (dc / dh carries, dMt, the filter-bank slabs)
// stays on plain accesses.  Spins are bounded (persist_sync.h) and raise the decode's device error word.
#include <stdlib.h>
#include <algorithm>
#include <vector>
#include "attention_bwd_body.h"
#include "skinny_body.h"
#include "persist_sync.h"

namespace {

constexpr int PB_WGS = 256;

struct PbwdArgs {
    AttnBwdArgs attn;                               // per-step pointers hold their value for step 0
    SkinnyArgs hcol, cell_a, ctxc;                  // likewise
    long BA, BD, BL, BH, B4H, Bp4H;                 // per-step strides (elements)
    float* ring; long slot, off_h;                  // ring slot t % ring_slots: [ctx slabs | h slabs]
    int ring_slots, T, a0, a1, ksb, ksc, n_attn, nch, zone, mask_a;
    float *dc_att, *dhc_att;
    PsSync sync;
};

__device__ __forceinline__ float* ring_at(const PbwdArgs& P, int t) { return P.ring + (long)(t % P.ring_slots) * P.slot; }

// The argument block is read from the kernarg segment AGAIN in every stage (scalar loads, scalar cache hits): with the block's ~700
// words visible as loop invariants the compiler hoists them out of the step loop and spills hundreds of SGPRs (and, through the lanes
// that hold them, VGPRs).  The pointer goes through an empty asm per stage, which ends every such live range.
typedef const PbwdArgs __attribute__((address_space(4)))* PbwdKernarg;
__device__ __forceinline__ const PbwdArgs& pb_args() {
    PbwdKernarg p = (PbwdKernarg)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return *(const PbwdArgs*)p;
}

__global__ __launch_bounds__(NT, 4) void pbwd_kernel(PbwdArgs P_unused) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float (&red4)[NW][64][17] = *reinterpret_cast<float (*)[NW][64][17]>(sm);
    float (&red1)[NW][16][17] = *reinterpret_cast<float (*)[NW][16][17]>(sm);
    const int id = blockIdx.x;
    unsigned epoch = 0;
    int n_it;
    {
        const PbwdArgs& P = pb_args();
        if (!ps_barrier(P.sync, ++epoch)) return;                  // start-up: every workgroup resident
        n_it = P.a1 - P.a0;
    }
    for (int i = 0; i < n_it; ++i) {
        // ---------------- stage 1: attention backward of step tA
        {
            const PbwdArgs& P = pb_args();
            const int tA = P.a1 - 1 - i;
            if (id < P.n_attn) {
                AttnBwdArgs q = P.attn;
                q.q += tA * P.BA; q.ctx += tA * P.BD; q.w += tA * P.BL; q.cum_in += tA * P.BL;
                if (q.dalign) q.dalign += tA * P.BL;
                q.dcum_out += tA * P.BL; q.dcum_in += tA * P.BL; q.dctx += tA * P.BD; q.dctx_total += tA * P.BD; q.dq += tA * P.BA;
                if (tA < P.T - 1) { q.part = ring_at(P, tA + 1); q.n_part = P.ksc; } else { q.part = q.dctx; q.n_part = 0; }
                attn_bwd_body(q, sm, id / P.nch, id % P.nch);
            }
        }
        {
            const PbwdArgs& P = pb_args();
            if (!ps_barrier(P.sync, ++epoch)) return;
        }
        // ---------------- stage 2: dq W_q + attention-LSTM cell backward of step tA -> dG_att(tA)
        {
            const PbwdArgs& P = pb_args();
            const int tA = P.a1 - 1 - i;
            const int cbs_h = P.cell_a.H >> 4, rt_cells = (P.cell_a.B + 15) >> 4;
            SkinnyArgs k = P.cell_a;
            k.seg[0].x += tA * P.BA; k.seg[1] = k.seg[0]; k.seg[2] = k.seg[0];
            k.dh_a += tA * P.BH;
            if (tA < P.T - 1) { k.part = ring_at(P, tA + 1) + P.off_h; k.n_part = P.ksb; } else { k.part = k.dh_a; k.n_part = 0; }
            k.gates += tA * P.B4H; k.c_prev += tA * P.BH;
            k.dc_in = P.dc_att + ((tA + 1) & 1) * P.BH; k.dc_out = P.dc_att + (tA & 1) * P.BH;
            if (P.zone) { k.dh_b = P.dhc_att + ((tA + 1) & 1) * P.BH; k.dh_carry_out = P.dhc_att + (tA & 1) * P.BH; }
            k.dgates_out += tA * P.B4H; k.dg_pack_out += tA * P.Bp4H;
            if (P.mask_a & 1) k.hmask += tA * P.BH;
            if (P.mask_a & 2) k.cmask += tA * P.BH;
            if (id < cbs_h * rt_cells) skinny_body<1, 1, 2, 0, 1>(k, red1, id % cbs_h, id / cbs_h, 0);
        }
        {
            const PbwdArgs& P = pb_args();
            if (!ps_barrier(P.sync, ++epoch)) return;
        }
        // ---------------- stage 3: ctx-columns of dG_att(tA) (what the next attention backward waits for) ...
        {
            const PbwdArgs& P = pb_args();
            const int tA = P.a1 - 1 - i;
            if (tA > 0) {
                const int cbs_c = P.ctxc.N >> 4;
                SkinnyArgs k = P.ctxc;
                k.seg[0].x += tA * P.Bp4H; k.seg[1] = k.seg[0]; k.seg[2] = k.seg[0];
                k.out = ring_at(P, tA);
                if (id < cbs_c * P.ksc) skinny_body<4, 2, 1, 1, 1>(k, red4, id % cbs_c, 0, id / cbs_c);
            }
        }
        {
            const PbwdArgs& P = pb_args();
            ps_bar_arrive(P.sync, ++epoch);
        }
        // ---------------- ... and, in the shadow of that barrier, the h-columns of dG_att(tA): only the cell backward of the NEXT iteration
        //                  reads them, and the barrier in front of it publishes them
        {
            const PbwdArgs& P = pb_args();
            const int tA = P.a1 - 1 - i;
            if (tA > 0) {
                const int cbs_h = P.cell_a.H >> 4;
                SkinnyArgs k = P.hcol;
                k.seg[0].x += tA * P.Bp4H; k.seg[1] = k.seg[0]; k.seg[2] = k.seg[0];
                k.out = ring_at(P, tA) + P.off_h;
                if (id < cbs_h * P.ksb) skinny_body<4, 2, 1, 1, 1>(k, red4, id % cbs_h, 0, id / cbs_h);
            }
        }
        {
            const PbwdArgs& P = pb_args();
            if (!ps_bar_wait(P.sync, epoch)) return;
        }
    }
}

size_t pbwd_lds(const AttnBwdArgs& q) {
    size_t lds = attn_bwd_fast_lds(q);
    const size_t lds_sk = sizeof(float) * NW * 64 * 17;
    return lds > lds_sk ? lds : lds_sk;
}

AttnBwdArgs attn_template(const DecoderArgs& a, const DecoderGradArgs& g) {
    const long BD = (long)a.B * a.Dm, BL = (long)a.B * a.L;
    AttnBwdArgs q; memset(&q, 0, sizeof(q));
    q.q = a.q_all; q.Mt = a.Mt; q.U = a.U; q.bias = a.att_bias; q.v = a.w_energy; q.memory = a.memory;
    q.ctx = a.ctx + BD; q.lengths = a.lengths; q.w = a.align; q.cum_in = a.cum;
    q.dalign = g.dalign;
    q.dcum_out = g.dcum_all + BL; q.dcum_in = g.dcum_all;
    q.dctx = g.dctx_all + BD; q.dctx_total = g.dctx_tot + BD;
    q.part_ks = BD; q.part_ld = a.Dm;
    q.dq = g.dq_all; q.dMt = g.dMt; q.dU_slab = g.dU_slab; q.dv_slab = g.dv_slab; q.dbias_slab = g.dbias_slab;
    q.B = a.B; q.L = a.L; q.A = a.A; q.Dm = a.Dm; q.ksz = a.ksz; q.nch = g.nch;
    q.n_part = g.ksb_ctx > 0 ? g.ksb_ctx : g.ksb;      // for the shape predicate
    return q;
}

struct Go { PbwdArgs* p; size_t lds; };
void pbwd_go(void* ctx, unsigned* cnt, unsigned* err, hipStream_t s) {
    Go* g = (Go*)ctx;
    g->p->sync.cnt = cnt; g->p->sync.err = err;
    hipLaunchKernelGGL(pbwd_kernel, dim3(PB_WGS), dim3(NT), g->lds, s, *g->p);
}

}  // namespace

// OFF by default: measured SLOWER than the per-step launch schedule on the bench shape (profiles/r06_pbwd_ab.txt: decoder backward
// 39.5-40.1 ms against 35.1-35.2 on one box; the in-kernel stage clock - MTTS_PBWD_CLOCK=1 - shows the bodies themselves running
// 1.2-2x longer inside the resident kernel than as their own launches: one workgroup per CU executes attention, cell and both
// products one after the other, where the launch schedule keeps two or three chain workgroups on a CU that cover each other's
// round trips).  MTTS_PBWD=1 selects it; tests/test_gpu_persist.py holds it to the per-step gradients.
static bool pbwd_enabled() {
    static const bool on = [] { const char* e = getenv("MTTS_PBWD"); return e && e[0] == '1'; }();
    return on && persist_enabled();
}

// slots of the partial-slab ring a chunk length needs (one slot per step of a launch + the slot the next launch reads)
static int pbwd_ring_slots(int chunk) { return chunk + 2; }

bool pbwd_supported(const DecoderArgs& a, const DecoderGradArgs& g) {
    const int ksc = g.ksb_ctx > 0 ? g.ksb_ctx : g.ksb;
    if (!(pbwd_enabled() && a.fast && a.precision == 0 && a.B > 32 && a.B <= 64 && (a.H & 15) == 0 && (a.Dm & 15) == 0 && (a.A & 3) == 0 &&
          a.persist_ws && a.q_all && a.gates_att && a.gates_gen && g.dG_att_p && g.dG_gen_p && g.att_w_rec_Tp && g.gen_w_hh_Tp &&
          g.part_ring && g.part_ring_slots >= pbwd_ring_slots(decoder_chunk()) && g.ksb >= 2 && g.ksb <= MAX_PART && ksc >= 2 && ksc <= BNP_MAX &&
          a.B * g.nch <= PB_WGS && (a.H >> 4) * ((a.B + 15) >> 4) <= PB_WGS && (a.H >> 4) * g.ksb <= PB_WGS && (a.Dm >> 4) * ksc <= PB_WGS))      // one tile per workgroup and stage
        return false;
    const AttnBwdArgs q = attn_template(a, g);
    if (!attn_bwd_fast_ok(q)) return false;
    return ps_ready_ext((const void*)pbwd_kernel, NT, pbwd_lds(q));
}

int pbwd_launch(const DecoderArgs& a, const DecoderGradArgs& g, const PbwdChunk& c, hipStream_t s) {
    MTTS_REQUIRE(pbwd_supported(a, g), "pbwd_launch: unsupported shape");
    if (c.a1 <= c.a0) return 0;
    const int B = a.B, H = a.H, A = a.A, Dm = a.Dm, T = a.T;
    const int ksb = g.ksb, ksc = g.ksb_ctx > 0 ? g.ksb_ctx : g.ksb;
    PbwdArgs P; memset(&P, 0, sizeof(P));
    P.BA = (long)B * A; P.BD = (long)B * Dm; P.BL = (long)B * a.L; P.BH = (long)B * H; P.B4H = 4 * P.BH; P.Bp4H = (long)((B + 15) & ~15) * 4 * H;
    P.ring = g.part_ring; P.ring_slots = g.part_ring_slots;
    P.off_h = (long)ksc * B * Dm; P.slot = P.off_h + (long)ksb * B * H;
    P.T = T; P.a0 = c.a0; P.a1 = c.a1; P.ksb = ksb; P.ksc = ksc; P.nch = g.nch; P.n_attn = B * g.nch;
    P.zone = a.zone ? 1 : 0;
    P.dc_att = g.dc_att; P.dhc_att = g.dh_carry_att;
    P.attn = attn_template(a, g);
    const int cb_ctx = Dm >> 4;
    auto product = [&](SkinnyArgs& k, const float* xp, const float* wp, int N, int ks, long out_ks) {
        memset(&k, 0, sizeof(k));
        k.nseg = 1; k.B = B; k.N = N; k.ksplit = ks;
        k.seg[0] = SkSeg{xp, wp, 4 * H, 4 * H, 4 * H, 1, 1};
        k.ldo = N; k.out_ks = out_ks;
    };
    product(P.hcol, g.dG_att_p, g.att_w_rec_Tp + (long)cb_ctx * (4 * H / 16) * 256, H, ksb, P.BH);
    product(P.ctxc, g.dG_att_p, g.att_w_rec_Tp, Dm, ksc, P.BD);
    auto cell = [&](SkinnyArgs& k, const float* gates, const float* c_prev, float* dG, float* dGp, const float* dh_a,
                    const uint8_t* hmask, const uint8_t* cmask, int& mask_bits) {
        memset(&k, 0, sizeof(k));
        k.B = B; k.H = H; k.N = H; k.lstm = 2; k.nseg = 1; k.ksplit = 1;
        k.dh_a = dh_a; k.ld_dh_a = H;
        k.part_ks = P.BH; k.part_ld = H; k.part_col0 = 0;
        k.gates = gates; k.c_prev = c_prev;
        k.dgates_out = dG; k.ld_dgates = 4 * H; k.dg_pack_out = dGp;
        mask_bits = 0;
        if (a.zone) { k.zone = 1; k.hmask = hmask; k.cmask = cmask; mask_bits = (hmask ? 1 : 0) | (cmask ? 2 : 0); }
        else if (a.training && hmask && a.p_hidden > 0.f) { k.zone = 0; k.hmask = hmask; k.hscale = 1.f / (1.f - a.p_hidden); mask_bits = 1; }
    };
    cell(P.cell_a, a.gates_att, a.c_att, g.dG_att, g.dG_att_p, g.dHA, a.att_hmask, a.att_cmask, P.mask_a);
    P.cell_a.seg[0] = SkSeg{g.dq_all, g.w_query_T, A, A, A, 0, 0};
    Go go{&P, pbwd_lds(P.attn)};
    MTTS_TRY(ps_run_launch(a, s, pbwd_go, &go));
    return 0;
}

MTTS_API int mtts_decoder_bwd_ring_slots(void) { return pbwd_ring_slots(decoder_chunk()); }
