// Back-propagation through time for the decoder loop and the encoder BiLSTM (host orchestration).
//
// The reference relies on autograd over 600 recorded steps (modules/tacotron2.py:180-198); here the sweep is
// explicit.  With every step teacher forced the two recurrences decouple:
//   chain B (generator LSTM) runs first: dh_gen_t = (batched projection part) + W_hh^T dG_{t+1}
//   chain A (attention LSTM + attention) afterwards, fed by the batched input gradients of chain B.
// Per step only skinny GEMMs against TRANSPOSED recurrent weights, the fused LSTM-cell backward epilogue and the
// attention backward kernel run; every weight gradient is one large MFMA GEMM over the saved gate gradients.
#include "common.h"
#include <chrono>
#include <stdlib.h>
#include <algorithm>
#include <vector>

static inline int round4(int x) { return (x + 3) & ~3; }

static int gm(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, bool tA, bool tB,
              float beta, hipStream_t s) {
    return gemm_plain(A, B, C, M, N, K, lda, ldb, ldc, tA, tB, 1.f, beta, nullptr, 0, s);
}

static void bwd_reg(const DecoderArgs& a, SkinnyArgs& k, const uint8_t* hmask, const uint8_t* cmask, int t) {
    const long off = (long)t * a.B * a.H;
    if (a.zone) { k.zone = 1; k.hmask = hmask ? hmask + off : nullptr; k.cmask = cmask ? cmask + off : nullptr; }
    else if (a.training && hmask && a.p_hidden > 0.f) { k.zone = 0; k.hmask = hmask + off; k.hscale = 1.f / (1.f - a.p_hidden); }
}

// Batched part after the sequential sweeps: prenet, every weight gradient, memory gradient, attention parameters.
// `prenet_chain_done`: the general schedule already pushed the gradient through the prenet step by step (dpren holds dz).
static int bwd_post(const DecoderArgs& a, const DecoderGradArgs& g, bool prenet_chain_done, bool gen_wgrad_done, bool att_wgrad_done,
                    hipStream_t s) {
    const int B = a.B, L = a.L, T = a.T, M = a.M, P = a.P, H = a.H, A = a.A, Dm = a.Dm;
    const int Mo = round4(M + 1), TB = T * B;
    const long BH = (long)B * H, BD = (long)B * Dm, BP = (long)B * P;
    const float* dout1 = g.dout + (long)B * Mo;
    // ---- prenet backward (batched over all frames)
    const int n = a.n_prenet;
    const float pscale = a.p_prenet > 0.f ? 1.f / (1.f - a.p_prenet) : 1.f;
    float* dp_last = g.dpren + (long)(n - 1) * T * BP;
    if (!prenet_chain_done) MTTS_TRY(gm(g.dG_att, a.att_w_ih, dp_last, TB, P, 4 * H, 4 * H, P + Dm, P, false, true, 0.f, s));
    for (int i = n - 1; i >= 0; --i) {
        float* dz = g.dpren + (long)i * T * BP;
        if (!prenet_chain_done) MTTS_TRY(relu_mask_bwd(dz, a.prenet_act[i], dz, (long)T * BP, a.prenet_mask[i] ? pscale : 1.f, s));
        const float* xin = i == 0 ? (prenet_chain_done ? g.frames_fed : a.frames_in) : a.prenet_act[i - 1];
        const int Kin = i == 0 ? M : P;
        MTTS_TRY(gm(dz, xin, g.d_prenet_w[i], P, Kin, TB, P, Kin, Kin, true, true, 0.f, s));
        MTTS_TRY(colsum(dz, g.d_prenet_b[i], TB, P, P, g.colsum_ws, s));
        if (i > 0 && !prenet_chain_done) MTTS_TRY(gm(dz, a.prenet_w[i], g.dpren + (long)(i - 1) * T * BP, TB, P, P, P, P, P, false, true, 0.f, s));
    }
    const float* pren = a.prenet_act[n - 1];

    // ---- LSTM weight gradients
    if (!att_wgrad_done) {
        MTTS_TRY(gm(g.dG_att, pren, g.d_att_w_ih, 4 * H, P, TB, 4 * H, P, P + Dm, true, true, 0.f, s));
        MTTS_TRY(gm(g.dG_att, a.ctx, g.d_att_w_ih + P, 4 * H, Dm, TB, 4 * H, Dm, P + Dm, true, true, 0.f, s));
        MTTS_TRY(gm(g.dG_att, a.h_att, g.d_att_w_hh, 4 * H, H, TB, 4 * H, H, H, true, true, 0.f, s));
    }
    MTTS_TRY(colsum(g.dG_att, g.d_att_b_ih, TB, 4 * H, 4 * H, g.colsum_ws, s));
    MTTS_CHECK_HIP(hipMemcpyAsync(g.d_att_b_hh, g.d_att_b_ih, (size_t)4 * H * sizeof(float), hipMemcpyDeviceToDevice, s));   // b_ih and b_hh enter the gates identically
    if (!gen_wgrad_done) {
        MTTS_TRY(gm(g.dG_gen, a.h_att + BH, g.d_gen_w_ih, 4 * H, H, TB, 4 * H, H, H + Dm, true, true, 0.f, s));
        MTTS_TRY(gm(g.dG_gen, a.ctx + BD, g.d_gen_w_ih + H, 4 * H, Dm, TB, 4 * H, Dm, H + Dm, true, true, 0.f, s));
        MTTS_TRY(gm(g.dG_gen, a.h_gen, g.d_gen_w_hh, 4 * H, H, TB, 4 * H, H, H, true, true, 0.f, s));
    }
    MTTS_TRY(colsum(g.dG_gen, g.d_gen_b_ih, TB, 4 * H, 4 * H, g.colsum_ws, s));
    MTTS_CHECK_HIP(hipMemcpyAsync(g.d_gen_b_hh, g.d_gen_b_ih, (size_t)4 * H * sizeof(float), hipMemcpyDeviceToDevice, s));

    // ---- frame/stop projection and query weights
    if (!att_wgrad_done) {
        MTTS_TRY(gm(dout1, a.h_gen + BH, g.d_w_out, M + 1, H, TB, Mo, H, H + Dm, true, true, 0.f, s));
        MTTS_TRY(gm(dout1, a.ctx + BD, g.d_w_out + H, M + 1, Dm, TB, Mo, Dm, H + Dm, true, true, 0.f, s));
        MTTS_TRY(gm(g.dq_all, a.h_att + BH, g.d_w_query, A, H, TB, A, H, H, true, true, 0.f, s));
    }
    MTTS_TRY(colsum(dout1, g.d_b_out, TB, M + 1, Mo, g.colsum_ws, s));

    // ---- memory gradient: context path (per-sample align^T dctx), memory-transform path, and W_memory
    {
        GemmArgs q; memset(&q, 0, sizeof(q));
        q.A = a.align; q.B = g.dctx_tot + BD; q.C = g.dmemory;
        q.M = L; q.N = Dm; q.K = T; q.lda = B * L; q.ldb = B * Dm; q.ldc = Dm;
        q.transA = 1; q.transB = 1; q.taps = 1; q.Kc = T; q.batch = B; q.zt = 1;
        q.a_z = L; q.b_z = Dm; q.c_z = (long)L * Dm; q.alpha = 1.f; q.mask_scale = 1.f;
        MTTS_TRY(mtts_gemm_ex(&q, s));
    }
    MTTS_TRY(gm(g.dMt, a.w_memory, g.dmemory, B * L, Dm, A, A, Dm, Dm, false, true, 1.f, s));
    MTTS_TRY(gm(g.dMt, a.memory, g.d_w_memory, A, Dm, B * L, A, Dm, Dm, true, true, 0.f, s));

    // ---- small attention parameters from the per-workgroup slabs
    const int nslab = B * g.nch;
    MTTS_TRY(colsum(g.dU_slab, g.dU, nslab, A * a.ksz, A * a.ksz, g.colsum_ws, s));
    MTTS_TRY(gm(g.dU, a.w_conv, g.d_w_loc, A, a.C, a.ksz, a.ksz, a.ksz, a.C, false, false, 0.f, s));
    MTTS_TRY(gm(a.w_loc, g.dU, g.d_w_conv, a.C, a.ksz, A, a.C, a.ksz, a.ksz, true, true, 0.f, s));
    MTTS_TRY(colsum(g.dv_slab, g.d_w_energy, nslab, A, A, g.colsum_ws, s));
    MTTS_TRY(colsum(g.dbias_slab, g.d_att_bias, nslab, A, A, g.colsum_ws, s));
    return 0;
}

// General schedule (some steps fed with the model's own prediction): one dependent chain per step, last step first.
//   projection bwd -> generator cell bwd -> generator input gradient -> attention bwd -> query+attention cell bwd
//   -> attention-LSTM input gradient -> prenet bwd (-> gradient of the previous frame when that step was free running)
static int bwd_general(const DecoderArgs& a, const DecoderGradArgs& g, hipStream_t s) {
    const int B = a.B, L = a.L, T = a.T, M = a.M, P = a.P, H = a.H, A = a.A, Dm = a.Dm, n = a.n_prenet;
    const int Mo = round4(M + 1);
    const long BH = (long)B * H, BD = (long)B * Dm, BL = (long)B * L, B4H = 4 * BH, BP = (long)B * P, BA = (long)B * A;
    MTTS_REQUIRE(g.att_w_ih_T && g.gen_w_ih_T && g.w_out_T && g.step_ws && g.frames_fed && (Mo & 3) == 0,
                 "decoder backward (general schedule) needs the transposed-weight and per-step workspaces");
    const int Fa = P + Dm + H, Fg = 2 * H + Dm, Fp = H + Dm;
    float* dfeat = g.step_ws;                    // [B, P+Dm+H]  gradient w.r.t. the attention-LSTM input of the LATER step
    float* dgen_in = dfeat + (long)B * Fa;       // [B, H+Dm+H]
    float* dproj = dgen_in + (long)B * Fg;       // [B, H+Dm]
    float* dframe = dproj + (long)B * Fp;        // [B, M]
    float* dout = const_cast<float*>(g.dout);
    const float pscale = a.p_prenet > 0.f ? 1.f / (1.f - a.p_prenet) : 1.f;

    // transposed weights
    MTTS_TRY(transpose2d(a.att_w_ih, g.att_w_ih_T, 4 * H, P + Dm, s));
    MTTS_TRY(transpose2d(a.att_w_hh, g.att_w_ih_T + (long)(P + Dm) * 4 * H, 4 * H, H, s));
    MTTS_TRY(transpose2d(a.gen_w_ih, g.gen_w_ih_T, 4 * H, H + Dm, s));
    MTTS_TRY(transpose2d(a.gen_w_hh, g.gen_w_ih_T + (long)(H + Dm) * 4 * H, 4 * H, H, s));
    MTTS_TRY(transpose2d(a.w_query, g.w_query_T, A, H, s));
    MTTS_CHECK_HIP(hipMemsetAsync(g.w_out_T, 0, (size_t)Fp * Mo * sizeof(float), s));
    MTTS_TRY(transpose2d_ld(a.w_out, g.w_out_T, M + 1, Fp, Fp, Mo, s));
    for (int i = 0; i < n; ++i) MTTS_TRY(transpose2d(a.prenet_w[i], g.prenet_w_T[i], P, i == 0 ? M : P, s));
    // frames actually fed: teacher frame or the model's previous output (slot t of `out` holds frame t-1)
    for (int t = 0; t < T; ++t) {
        const bool teach = a.frames_in && a.teacher && a.teacher[t];
        if (teach) MTTS_TRY(copy2d(a.frames_in + (long)t * B * M, g.frames_fed + (long)t * B * M, B, M, M, M, s));
        else MTTS_TRY(copy2d(a.out + (long)t * B * Mo, g.frames_fed + (long)t * B * M, B, M, Mo, M, s));
    }

    for (int t = T - 1; t >= 0; --t) {
        const bool last = t == T - 1;
        {   // projection backward: dproj = dout[t] W_out
            SkinnyArgs k; memset(&k, 0, sizeof(k));
            k.nseg = 1; k.B = B; k.N = Fp; k.ksplit = 1;
            k.seg[0] = SkSeg{dout + (long)(t + 1) * B * Mo, g.w_out_T, Mo, Mo, Mo, 0, 0};
            k.out = dproj; k.ldo = Fp;
            MTTS_TRY(skinny_launch(k, s));
        }
        {   // generator cell backward
            SkinnyArgs k; memset(&k, 0, sizeof(k));
            k.B = B; k.H = H; k.lstm = 2; k.nseg = 0; k.ksplit = 1;
            k.dh_a = dproj; k.ld_dh_a = Fp;
            if (!last) { k.part = dgen_in; k.n_part = 1; k.part_ks = 0; k.part_ld = Fg; k.part_col0 = H + Dm; }
            k.gates = a.gates_gen + t * B4H; k.c_prev = a.c_gen + t * BH;
            k.dc_in = g.dc_gen + ((t + 1) & 1) * BH; k.dc_out = g.dc_gen + (t & 1) * BH;
            if (a.zone) { k.dh_b = g.dh_carry_gen + ((t + 1) & 1) * BH; k.dh_carry_out = g.dh_carry_gen + (t & 1) * BH; }
            k.dgates_out = g.dG_gen + t * B4H; k.ld_dgates = 4 * H;
            bwd_reg(a, k, a.gen_hmask, a.gen_cmask, t);
            MTTS_TRY(skinny_launch(k, s));
        }
        {   // d[h_att_t, ctx_t, h_gen_{t-1}] = dG_gen [W_ih | W_hh]
            SkinnyArgs k; memset(&k, 0, sizeof(k));
            k.nseg = 1; k.B = B; k.N = Fg; k.ksplit = 1;
            k.seg[0] = SkSeg{g.dG_gen + t * B4H, g.gen_w_ih_T, 4 * H, 4 * H, 4 * H, 0, 0};
            k.out = dgen_in; k.ldo = Fg;
            MTTS_TRY(skinny_launch(k, s));
        }
        // total context gradient of this step: projection + generator input + attention-LSTM input of step t+1
        MTTS_TRY(add3(g.dctx_all + (t + 1) * BD, Dm, dproj + H, Fp, dgen_in + H, Fg, last ? nullptr : dfeat + P, Fa, B, Dm, false, s));
        {
            AttnBwdArgs q; memset(&q, 0, sizeof(q));
            q.q = a.q_all + t * BA; q.Mt = a.Mt; q.U = a.U; q.bias = a.att_bias; q.v = a.w_energy; q.memory = a.memory;
            q.ctx = a.ctx + (t + 1) * BD; q.lengths = a.lengths; q.w = a.align + t * BL; q.cum_in = a.cum + t * BL;
            q.dalign = g.dalign ? g.dalign + t * BL : nullptr;
            q.dcum_out = g.dcum_all + (t + 1) * BL; q.dcum_in = g.dcum_all + t * BL;
            q.dctx = g.dctx_all + (t + 1) * BD; q.dctx_total = g.dctx_tot + (t + 1) * BD;
            q.dq = g.dq_all + t * BA; q.dMt = g.dMt; q.dU_slab = g.dU_slab; q.dv_slab = g.dv_slab; q.dbias_slab = g.dbias_slab;
            q.B = B; q.L = L; q.A = A; q.Dm = Dm; q.ksz = a.ksz; q.nch = g.nch;
            MTTS_TRY(mtts_attn_step_bwd(&q, s));
        }
        {   // dh_att_t = dq W_q + (generator input part) + (recurrent part of step t+1) -> attention cell backward
            SkinnyArgs k; memset(&k, 0, sizeof(k));
            k.B = B; k.H = H; k.lstm = 2; k.nseg = 1; k.ksplit = 1;
            k.seg[0] = SkSeg{g.dq_all + t * BA, g.w_query_T, A, A, A, 0, 0};
            k.dh_a = dgen_in; k.ld_dh_a = Fg;
            if (!last) { k.part = dfeat; k.n_part = 1; k.part_ks = 0; k.part_ld = Fa; k.part_col0 = P + Dm; }
            k.gates = a.gates_att + t * B4H; k.c_prev = a.c_att + t * BH;
            k.dc_in = g.dc_att + ((t + 1) & 1) * BH; k.dc_out = g.dc_att + (t & 1) * BH;
            if (a.zone) { k.dh_b = g.dh_carry_att + ((t + 1) & 1) * BH; k.dh_carry_out = g.dh_carry_att + (t & 1) * BH; }
            k.dgates_out = g.dG_att + t * B4H; k.ld_dgates = 4 * H;
            bwd_reg(a, k, a.att_hmask, a.att_cmask, t);
            MTTS_TRY(skinny_launch(k, s));
        }
        {   // d[prenet_t, ctx_{t-1}, h_att_{t-1}] = dG_att [W_ih | W_hh]
            SkinnyArgs k; memset(&k, 0, sizeof(k));
            k.nseg = 1; k.B = B; k.N = Fa; k.ksplit = 1;
            k.seg[0] = SkSeg{g.dG_att + t * B4H, g.att_w_ih_T, 4 * H, 4 * H, 4 * H, 0, 0};
            k.out = dfeat; k.ldo = Fa;
            MTTS_TRY(skinny_launch(k, s));
        }
        // prenet backward for this step; the gradient reaches frame t-1 only when step t was free running
        const bool teach = a.frames_in && a.teacher && a.teacher[t];
        MTTS_TRY(copy2d(dfeat, g.dpren + ((long)(n - 1) * T + t) * BP, B, P, Fa, P, s));
        for (int i = n - 1; i >= 0; --i) {
            float* dz = g.dpren + ((long)i * T + t) * BP;
            MTTS_TRY(relu_mask_bwd(dz, a.prenet_act[i] + t * BP, dz, BP, a.prenet_mask[i] ? pscale : 1.f, s));
            if (i > 0 || (!teach && t > 0)) {
                SkinnyArgs k; memset(&k, 0, sizeof(k));
                k.nseg = 1; k.B = B; k.ksplit = 1;
                k.seg[0] = SkSeg{dz, g.prenet_w_T[i], P, P, P, 0, 0};
                if (i > 0) { k.N = P; k.out = g.dpren + ((long)(i - 1) * T + t) * BP; k.ldo = P; }
                else { k.N = M; k.out = dframe; k.ldo = M; }
                MTTS_TRY(skinny_launch(k, s));
            }
        }
        if (!teach && t > 0) MTTS_TRY(add3(dout + (long)t * B * Mo, Mo, dframe, M, nullptr, 0, nullptr, 0, B, M, true, s));
    }
    return bwd_post(a, g, true, false, false, s);
}

MTTS_API int mtts_decoder_bwd(const DecoderArgs* fwd, const DecoderGradArgs* grad, void* stream) {
    const DecoderArgs& a = *fwd;
    const DecoderGradArgs& g = *grad;
    hipStream_t s = (hipStream_t)stream;
    const int B = a.B, L = a.L, T = a.T, M = a.M, P = a.P, H = a.H, A = a.A, Dm = a.Dm;
    const int Mo = round4(M + 1), TB = T * B;
    const long BH = (long)B * H, BD = (long)B * Dm, BL = (long)B * L, B4H = 4 * BH, BP = (long)B * P, BA = (long)B * A;
    if (!a.fast) return bwd_general(a, g, s);
    MTTS_REQUIRE(!a.zone || (a.training && g.dh_carry_att && g.dh_carry_gen), "decoder backward with zoneout needs training mode and the carry buffers");
    MTTS_REQUIRE(a.q_all && a.gates_att && a.gates_gen, "decoder backward needs q_all and the saved gates");
    MTTS_REQUIRE((A & 3) == 0, "attention dimension must be a multiple of 4");
    const int ksb = g.ksb;

    // ---- transposed recurrent weights
    MTTS_TRY(transpose2d_ld(a.att_w_ih + P, g.att_w_rec_T, 4 * H, Dm, P + Dm, 4 * H, s));
    MTTS_TRY(transpose2d(a.att_w_hh, g.att_w_rec_T + (long)Dm * 4 * H, 4 * H, H, s));
    MTTS_TRY(transpose2d(a.gen_w_hh, g.gen_w_hh_T, 4 * H, H, s));
    MTTS_TRY(transpose2d(a.w_query, g.w_query_T, A, H, s));
    // bf16 mode (round 5): the per-step input-gradient products dG W^T of both chains run on bf16 MFMA - transposed recurrent weights
    // and the cell backward's packed copy of dG as RNE-rounded bf16 pair tiles (skinny_body.h, PK = 3): half the 42 MB of weights the
    // three products re-stream per step, 1/4 of their MFMA issue; fp32 accumulation, the cell / attention backward stay fp32
    const bool bfp = a.precision == 1 && g.dG_att_p && g.dG_gen_p && g.att_w_rec_Tp && g.gen_w_hh_Tp && (Dm & 15) == 0 && (H & 15) == 0;      // H % 16: the cell backward's pair-tile store derives lane and element from u alone (skinny_body.h bwd_cell)
    const int pkv = bfp ? 2 : 1;
    if (bfp) {
        MTTS_TRY(mtts_pack_weight_bf16(g.att_w_rec_T, 4 * H, Dm + H, 4 * H, g.att_w_rec_Tp, s));
        MTTS_TRY(mtts_pack_weight_bf16(g.gen_w_hh_T, 4 * H, H, 4 * H, g.gen_w_hh_Tp, s));
    } else {
        if (g.att_w_rec_Tp) MTTS_TRY(mtts_pack_weight(g.att_w_rec_T, 4 * H, Dm + H, 4 * H, 0, g.att_w_rec_Tp, s));
        if (g.gen_w_hh_Tp) MTTS_TRY(mtts_pack_weight(g.gen_w_hh_T, 4 * H, H, 4 * H, 0, g.gen_w_hh_Tp, s));
    }
    const long Bp4H = (long)((B + 15) & ~15) * 4 * H;
    const float* dout1 = g.dout + (long)B * Mo;      // slot 1 = step 0
    // ---- projection backward (batched): dHG = dout W_out[:, :H],  dctx_all[1:] = dout W_out[:, H:]
    MTTS_TRY(gm(dout1, a.w_out, g.dHG, TB, H, M + 1, Mo, H + Dm, H, false, true, 0.f, s));
    MTTS_TRY(gm(dout1, a.w_out + H, g.dctx_all + BD, TB, Dm, M + 1, Mo, H + Dm, Dm, false, true, 0.f, s));

    // ---- chain B (generator LSTM, side stream) runs AHEAD of chain A (attention + attention LSTM, caller's stream),
    //      chunk by chunk from the last step backwards; per chunk: recurrence -> batched input gradients -> event.
    const int CH = decoder_chunk();
    hipStream_t sb = side_stream(s);
    if (!sb) return mtts_fail("decoder backward: cannot create the side stream");
    {
        hipEvent_t ev = pool_event(s);
        MTTS_CHECK_HIP(hipEventRecord(ev, s));
        MTTS_CHECK_HIP(hipStreamWaitEvent(sb, ev, 0));
    }
    // Chunk bounds (ascending; chunk c = steps [bounds[c], bounds[c + 1]), processed from the last one down): uniform chunks of CH steps, the
    // ragged one at the end of the decode.  (Round 6 measured a ramp - 8, 16, 32 steps first so that chain A starts after 8 steps of chain B
    // instead of 48, a 16-step chunk last so that fewer weight-gradient GEMMs trail chain A: no difference, profiles/r06_chunk_ramp_ab.txt.)
    std::vector<int> bounds;
    for (int t = 0; t < T; t += CH) bounds.push_back(t);
    bounds.push_back(T);
    const int nchunks = (int)bounds.size() - 1;
    std::vector<hipEvent_t> chunk_ev(nchunks);
    // ---- weight gradients ride a third, least-priority stream: every chunk's dG^T x product is queued as soon as its chain
    //      has produced the chunk's dG, and accumulates (beta = 1) into the gradient; one workgroup per CU (nosplit) so that
    //      the step kernels of both chains always find room.  The frame-projection gradient needs no chain at all.
    //      Round-3 A/B (same box, ms per train step): beside the chains 84.0, deferred to the end of the chains on the caller's
    //      stream 88.3-89.1, helper stream confined to 64 / 128 / 192 CUs by hipExtStreamCreateWithCUMask 106-108 (not kept).
    hipStream_t sw = wgrad_stream(s);
    if (!sw) return mtts_fail("decoder backward: cannot create the weight-gradient stream");
    auto wgrad = [&](const float* dY, int ldy, int Mw, const float* X, int ldx, int Nw, float* dW, int ldw, int rows, float beta) -> int {
        GemmArgs q; memset(&q, 0, sizeof(q));
        q.taps = 1; q.batch = 1; q.zt = 1; q.alpha = 1.f; q.mask_scale = 1.f; q.nosplit = 2; q.transA = 1; q.transB = 1;
        q.A = dY; q.lda = ldy; q.M = Mw; q.B = X; q.ldb = ldx; q.N = Nw; q.C = dW; q.ldc = ldw; q.K = rows; q.Kc = rows; q.beta = beta;
        return mtts_gemm_ex(&q, sw);
    };
    {
        hipEvent_t ev = pool_event(s);
        MTTS_CHECK_HIP(hipEventRecord(ev, s));
        MTTS_CHECK_HIP(hipStreamWaitEvent(sw, ev, 0));
        MTTS_TRY(wgrad(dout1, Mo, M + 1, a.h_gen + BH, H, H, g.d_w_out, H + Dm, TB, 0.f));
        MTTS_TRY(wgrad(dout1, Mo, M + 1, a.ctx + BD, Dm, Dm, g.d_w_out + H, H + Dm, TB, 0.f));
    }
    static const bool host_timing = [] { const char* e = getenv("MTTS_HOST_TIMING"); return e && e[0] == '1'; }();
    const auto host_t0 = std::chrono::steady_clock::now();
    // MTTS_HOST_TIMING=1 (diagnosis): host submission time and a per-chunk GPU timeline of the three streams from timing events
    std::vector<hipEvent_t> tev;
    std::vector<const char*> tev_name;
    std::vector<int> tev_chunk;
    auto tmark = [&](hipStream_t st, const char* name, int c) {
        if (!host_timing) return;
        hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return;
        (void)hipEventRecord(e, st); tev.push_back(e); tev_name.push_back(name); tev_chunk.push_back(c);
    };
    tmark(s, "start", -1);
    const float* pren = a.prenet_act[a.n_prenet - 1];
    auto submit_B = [&](int c) -> int {
        const int c0 = bounds[c], c1 = bounds[c + 1], n = c1 - c0;
        for (int t = c1 - 1; t >= c0; --t) {
            SkinnyArgs k; memset(&k, 0, sizeof(k));
            k.B = B; k.H = H; k.lstm = 2; k.nseg = 0; k.ksplit = 1;
            k.dh_a = g.dHG + t * BH; k.ld_dh_a = H;
            if (t < T - 1) { k.part = g.part_gen; k.n_part = ksb; k.part_ks = BH; k.part_ld = H; k.part_col0 = 0; }
            k.gates = a.gates_gen + t * B4H; k.c_prev = a.c_gen + t * BH;
            k.dc_in = g.dc_gen + ((t + 1) & 1) * BH; k.dc_out = g.dc_gen + (t & 1) * BH;
            if (a.zone) { k.dh_b = g.dh_carry_gen + ((t + 1) & 1) * BH; k.dh_carry_out = g.dh_carry_gen + (t & 1) * BH; }
            k.dgates_out = g.dG_gen + t * B4H; k.ld_dgates = 4 * H;
            k.dg_pack_out = g.dG_gen_p ? g.dG_gen_p + t * Bp4H : nullptr;
            k.dg_pack_bf16 = bfp ? 1 : 0;
            bwd_reg(a, k, a.gen_hmask, a.gen_cmask, t);
            MTTS_TRY(skinny_launch(k, sb));
            if (t > 0) {
                SkinnyArgs q; memset(&q, 0, sizeof(q));
                q.nseg = 1; q.B = B; q.N = H; q.ksplit = ksb;
                q.seg[0] = SkSeg{g.dG_gen + t * B4H, g.gen_w_hh_T, 4 * H, 4 * H, 4 * H, 0, 0};
                if (g.dG_gen_p) { q.seg[0].x = g.dG_gen_p + t * Bp4H; q.seg[0].xpack = pkv; }
                if (g.gen_w_hh_Tp) { q.seg[0].w = g.gen_w_hh_Tp; q.seg[0].wpack = pkv; }
                q.out = g.part_gen; q.ldo = H; q.out_ks = BH;
                MTTS_TRY(skinny_launch(q, sb));
            }
        }
        // input gradients of the generator LSTM for this chunk: dHA = dG_gen W_ih[:, :H];  dctx_all[1:] += dG_gen W_ih[:, H:]
        GemmArgs q; memset(&q, 0, sizeof(q));
        q.taps = 1; q.batch = 1; q.zt = 1; q.alpha = 1.f; q.mask_scale = 1.f; q.nosplit = 1; q.transB = 1;
        q.A = g.dG_gen + c0 * B4H; q.B = a.gen_w_ih; q.C = g.dHA + c0 * BH;
        q.M = n * B; q.N = H; q.K = 4 * H; q.Kc = 4 * H; q.lda = 4 * H; q.ldb = H + Dm; q.ldc = H; q.beta = 0.f;
        MTTS_TRY(mtts_gemm_ex(&q, sb));
        q.B = a.gen_w_ih + H; q.C = g.dctx_all + (c0 + 1) * BD; q.N = Dm; q.ldc = Dm; q.beta = 1.f;
        MTTS_TRY(mtts_gemm_ex(&q, sb));
        chunk_ev[c] = pool_event(s);
        MTTS_CHECK_HIP(hipEventRecord(chunk_ev[c], sb));
        tmark(sb, "B end", c);
        // generator-LSTM weight gradients of this chunk
        MTTS_CHECK_HIP(hipStreamWaitEvent(sw, chunk_ev[c], 0));
        const float beta = c == nchunks - 1 ? 0.f : 1.f;
        MTTS_TRY(wgrad(g.dG_gen + c0 * B4H, 4 * H, 4 * H, a.h_att + (c0 + 1) * BH, H, H, g.d_gen_w_ih, H + Dm, n * B, beta));
        MTTS_TRY(wgrad(g.dG_gen + c0 * B4H, 4 * H, 4 * H, a.ctx + (c0 + 1) * BD, Dm, Dm, g.d_gen_w_ih + H, H + Dm, n * B, beta));
        MTTS_TRY(wgrad(g.dG_gen + c0 * B4H, 4 * H, 4 * H, a.h_gen + c0 * BH, H, H, g.d_gen_w_hh, H, n * B, beta));
        tmark(sw, "Wgen end", c);
        return 0;
    };
    // Chain A per step t:  { attention backward(t)  ||  dG_att(t+1) W_hh^T }  ->  dq W_q + cell backward(t)  ->  dG_att(t) W_ih[:, P:]^T
    // The h-columns of the input gradient are only needed by the cell backward, so they share the attention backward's launch
    // ("fat launch"); the ctx-columns sit on the critical path and get their own, smaller launch.
    const int ksc = g.ksb_ctx > 0 ? g.ksb_ctx : ksb;
    float* part_ctx = g.part_att;                               // [ksc][B][Dm]
    float* part_h = g.part_att + (long)ksc * B * Dm;            // [ksb][B][H]
    const int cb_ctx = Dm >> 4;                                 // packed tiles of the ctx rows of [W_ih[:, P:] | W_hh]^T
    auto gemm_seg = [&](int t, bool h_part) {
        SkSeg sg = SkSeg{g.dG_att + t * B4H, g.att_w_rec_T + (h_part ? (long)Dm * 4 * H : 0), 4 * H, 4 * H, 4 * H, 0, 0};
        if (g.dG_att_p) { sg.x = g.dG_att_p + t * Bp4H; sg.xpack = pkv; }
        if (g.att_w_rec_Tp && (Dm & 15) == 0) { sg.w = g.att_w_rec_Tp + (h_part ? (long)cb_ctx * (4 * H / (bfp ? 32 : 16)) * 256 : 0); sg.wpack = pkv; }
        return sg;
    };
    const bool persistent_a = pbwd_supported(a, g);
    auto submit_A = [&](int c) -> int {
        const int c0 = bounds[c], c1 = bounds[c + 1];
        MTTS_CHECK_HIP(hipStreamWaitEvent(s, chunk_ev[c], 0));
        tmark(s, "A start", c);
        if (persistent_a) {      // round 6 (csrc/pbwd.hip): the chunk's steps of chain A in ONE resident launch
            MTTS_TRY(pbwd_launch(a, g, PbwdChunk{c0, c1}, s));
        } else
        for (int t = c1 - 1; t >= c0; --t) {
            {
                AttnBwdArgs q; memset(&q, 0, sizeof(q));
                q.q = a.q_all + t * BA; q.Mt = a.Mt; q.U = a.U; q.bias = a.att_bias; q.v = a.w_energy; q.memory = a.memory;
                q.ctx = a.ctx + (t + 1) * BD; q.lengths = a.lengths; q.w = a.align + t * BL; q.cum_in = a.cum + t * BL;
                q.dalign = g.dalign ? g.dalign + t * BL : nullptr;
                q.dcum_out = g.dcum_all + (t + 1) * BL; q.dcum_in = g.dcum_all + t * BL;
                q.dctx = g.dctx_all + (t + 1) * BD; q.dctx_total = g.dctx_tot + (t + 1) * BD;
                if (t < T - 1) { q.part = part_ctx; q.n_part = ksc; q.part_ks = BD; q.part_ld = Dm; }
                q.dq = g.dq_all + t * BA; q.dMt = g.dMt; q.dU_slab = g.dU_slab; q.dv_slab = g.dv_slab; q.dbias_slab = g.dbias_slab;
                q.B = B; q.L = L; q.A = A; q.Dm = Dm; q.ksz = a.ksz; q.nch = g.nch;
                if (t < T - 1) {     // fat launch with the h-columns of step t+1's input gradient
                    SkinnyArgs k; memset(&k, 0, sizeof(k));
                    k.nseg = 1; k.B = B; k.N = H; k.ksplit = ksb;
                    k.seg[0] = gemm_seg(t + 1, true);
                    k.out = part_h; k.ldo = H; k.out_ks = BH;
                    MTTS_TRY(attn_bwd_plus_skinny(q, k, s));
                } else {
                    MTTS_TRY(mtts_attn_step_bwd(&q, s));
                }
            }
            {   // dh_att_t = dq W_q + dHA[t] + (recurrent part of step t+1) -> cell backward
                SkinnyArgs k; memset(&k, 0, sizeof(k));
                k.B = B; k.H = H; k.lstm = 2; k.nseg = 1; k.ksplit = 1;
                k.seg[0] = SkSeg{g.dq_all + t * BA, g.w_query_T, A, A, A, 0, 0};
                k.dh_a = g.dHA + t * BH; k.ld_dh_a = H;
                if (t < T - 1) { k.part = part_h; k.n_part = ksb; k.part_ks = BH; k.part_ld = H; k.part_col0 = 0; }
                k.gates = a.gates_att + t * B4H; k.c_prev = a.c_att + t * BH;
                k.dc_in = g.dc_att + ((t + 1) & 1) * BH; k.dc_out = g.dc_att + (t & 1) * BH;
                if (a.zone) { k.dh_b = g.dh_carry_att + ((t + 1) & 1) * BH; k.dh_carry_out = g.dh_carry_att + (t & 1) * BH; }
                k.dgates_out = g.dG_att + t * B4H; k.ld_dgates = 4 * H;
                k.dg_pack_out = g.dG_att_p ? g.dG_att_p + t * Bp4H : nullptr;
                k.dg_pack_bf16 = bfp ? 1 : 0;
                bwd_reg(a, k, a.att_hmask, a.att_cmask, t);
                MTTS_TRY(skinny_launch(k, s));
            }
            if (t > 0) {   // d ctx_{t-1} = dG_att_t W_ih[:, P:]  (critical path of the next attention backward)
                SkinnyArgs q; memset(&q, 0, sizeof(q));
                q.nseg = 1; q.B = B; q.N = Dm; q.ksplit = ksc;
                q.seg[0] = gemm_seg(t, false);
                q.out = part_ctx; q.ldo = Dm; q.out_ks = BD;
                MTTS_TRY(skinny_launch(q, s));
            }
        }
        tmark(s, "A end", c);
        {   // attention-LSTM and query weight gradients of this chunk
            hipEvent_t ev = pool_event(s);
            MTTS_CHECK_HIP(hipEventRecord(ev, s));
            MTTS_CHECK_HIP(hipStreamWaitEvent(sw, ev, 0));
            const int n = c1 - c0;
            const float beta = c == nchunks - 1 ? 0.f : 1.f;
            const float* dG = g.dG_att + c0 * B4H;
            MTTS_TRY(wgrad(dG, 4 * H, 4 * H, pren + c0 * BP, P, P, g.d_att_w_ih, P + Dm, n * B, beta));
            MTTS_TRY(wgrad(dG, 4 * H, 4 * H, a.ctx + c0 * BD, Dm, Dm, g.d_att_w_ih + P, P + Dm, n * B, beta));
            MTTS_TRY(wgrad(dG, 4 * H, 4 * H, a.h_att + c0 * BH, H, H, g.d_att_w_hh, H, n * B, beta));
            MTTS_TRY(wgrad(g.dq_all + c0 * BA, A, A, a.h_att + (c0 + 1) * BH, H, H, g.d_w_query, H, n * B, beta));
            tmark(sw, "Watt end", c);
        }
        return 0;
    };
    // host submission order keeps chain B one chunk ahead of chain A so that neither stream starves
    MTTS_TRY(submit_B(nchunks - 1));
    for (int c = nchunks - 1; c >= 0; --c) {
        if (c > 0) MTTS_TRY(submit_B(c - 1));
        MTTS_TRY(submit_A(c));
    }
    if (host_timing)
        fprintf(stderr, "[mtts] decoder_bwd: chains submitted in %.3f ms of host time (T=%d)\n",
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - host_t0).count(), T);
    if (host_timing && !tev.empty()) {
        (void)hipStreamSynchronize(sb); (void)hipStreamSynchronize(sw); (void)hipStreamSynchronize(s);
        for (size_t i = 1; i < tev.size(); ++i) {
            float ms = 0.f; (void)hipEventElapsedTime(&ms, tev[0], tev[i]);
            fprintf(stderr, "[mtts]   %-9s chunk %2d  %8.3f ms\n", tev_name[i], tev_chunk[i], ms);
        }
        for (hipEvent_t e : tev) (void)hipEventDestroy(e);
    }
    hipEvent_t ev_b_done = pool_event(s), ev_w_done = pool_event(s);
    MTTS_CHECK_HIP(hipEventRecord(ev_b_done, sb));
    MTTS_CHECK_HIP(hipEventRecord(ev_w_done, sw));
    MTTS_CHECK_HIP(hipStreamWaitEvent(s, ev_b_done, 0));
    MTTS_CHECK_HIP(hipStreamWaitEvent(s, ev_w_done, 0));     // join before the split-K GEMMs of the batched part (shared scratch)

    return bwd_post(a, g, false, true, true, s);
}

MTTS_API int mtts_bilstm_bwd(const BiLstmArgs* fwd, const BiLstmGradArgs* grad, void* stream) {
    const BiLstmArgs& a = *fwd;
    const BiLstmGradArgs& g = *grad;
    hipStream_t s = (hipStream_t)stream;
    const int B = a.B, L = a.L, H = a.H, Cin = a.Cin;
    const long BH = (long)B * H, B4H = 4 * BH;
    // both directions' recurrences run concurrently (reverse direction on the side stream, own scratch halves); the batched
    // products that follow share dx and the split-K scratch and stay on the caller's stream
    hipStream_t sd[2] = {s, side_stream(s)};
    if (!sd[1]) return mtts_fail("bilstm backward: cannot create the side stream");
    hipEvent_t ev_fork = pool_event(s), ev_join = pool_event(s);
    MTTS_CHECK_HIP(hipEventRecord(ev_fork, s));
    MTTS_CHECK_HIP(hipStreamWaitEvent(sd[1], ev_fork, 0));
    for (int d = 1; d >= 0; --d) {
        MTTS_REQUIRE(a.gates[d], "bilstm backward needs the saved gates");
        hipStream_t q_s = sd[d];
        float* part = g.part + (long)d * g.ksb * BH;
        float* dc = g.dc + (long)d * 2 * BH;
        float* dh_carry = g.dh_carry + (long)d * 2 * BH;
        MTTS_TRY(transpose2d(a.w_hh[d], g.w_hh_T[d], 4 * H, H, q_s));
        MTTS_CHECK_HIP(hipMemsetAsync(dc, 0, 2 * BH * sizeof(float), q_s));
        MTTS_CHECK_HIP(hipMemsetAsync(dh_carry, 0, 2 * BH * sizeof(float), q_s));
        for (int st = L - 1; st >= 0; --st) {      // reverse of the forward processing order
            const int t = d == 0 ? st : L - 1 - st;
            const int s_in = d == 0 ? t : t + 1;
            SkinnyArgs k; memset(&k, 0, sizeof(k));
            k.B = B; k.H = H; k.lstm = 2; k.nseg = 0; k.ksplit = 1;
            k.dh_a = g.dy + (long)t * 2 * H + d * H; k.ld_dh_a = L * 2 * H;
            k.dh_b = dh_carry + ((st + 1) & 1) * BH; k.dh_carry_out = dh_carry + (st & 1) * BH;
            if (st < L - 1) { k.part = part; k.n_part = g.ksb; k.part_ks = BH; k.part_ld = H; k.part_col0 = 0; }
            k.gates = a.gates[d] + (long)t * B4H; k.c_prev = a.c[d] + s_in * BH;
            k.dc_in = dc + ((st + 1) & 1) * BH; k.dc_out = dc + (st & 1) * BH;
            k.lengths = a.lengths; k.t = t;
            k.dgates_out = g.dxproj[d] + (long)t * B4H; k.ld_dgates = 4 * H;
            MTTS_TRY(skinny_launch(k, q_s));
            if (st > 0) {
                SkinnyArgs q; memset(&q, 0, sizeof(q));
                q.nseg = 1; q.B = B; q.N = H; q.ksplit = g.ksb;
                q.seg[0] = SkSeg{g.dxproj[d] + (long)t * B4H, g.w_hh_T[d], 4 * H, 4 * H, 4 * H, 0, 0};
                q.out = part; q.ldo = H; q.out_ks = BH;
                MTTS_TRY(skinny_launch(q, q_s));
            }
        }
    }
    MTTS_CHECK_HIP(hipEventRecord(ev_join, sd[1]));
    MTTS_CHECK_HIP(hipStreamWaitEvent(s, ev_join, 0));
    for (int d = 0; d < 2; ++d) {
        // dx (+)= dG W_ih ; dW_ih = dG^T x ; dW_hh = dG^T h_prev ; biases
        MTTS_TRY(gm(g.dxproj[d], a.w_ih[d], g.dx, L * B, Cin, 4 * H, 4 * H, Cin, Cin, false, true, d == 0 ? 0.f : 1.f, s));
        MTTS_TRY(gm(g.dxproj[d], a.x, g.d_w_ih[d], 4 * H, Cin, L * B, 4 * H, Cin, Cin, true, true, 0.f, s));
        const float* hprev = d == 0 ? a.h[d] : a.h[d] + BH;
        MTTS_TRY(gm(g.dxproj[d], hprev, g.d_w_hh[d], 4 * H, H, L * B, 4 * H, H, H, true, true, 0.f, s));
        MTTS_TRY(colsum(g.dxproj[d], g.d_b_ih[d], L * B, 4 * H, 4 * H, g.colsum_ws, s));
        MTTS_CHECK_HIP(hipMemcpyAsync(g.d_b_hh[d], g.d_b_ih[d], (size_t)4 * H * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    return 0;
}
