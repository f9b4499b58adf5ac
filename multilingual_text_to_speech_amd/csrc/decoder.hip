// Host orchestration of the autoregressive decoder loop and the encoder BiLSTM (forward).
// Reference: Decoder._decode modules/tacotron2.py:148-209; Encoder BiLSTM modules/encoder.py:41-44.
//
// Two schedules over the same kernels:
//   general  - every step runs prenet(step) | att-LSTM | query | attention | gen-LSTM | frame/stop projection;
//              needed whenever a step consumes the model's own previous frame (inference, teacher forcing < 1).
//   fast     - all steps teacher forced: everything that does not sit on the recurrence is hoisted into large
//              MFMA GEMMs (prenet over all frames, att-LSTM input projection, gen-LSTM input projection,
//              frame/stop projection); the sequential part shrinks to
//              chain A: att-LSTM([ctx,h]) -> query -> attention      (per step)
//              chain B: gen-LSTM(h_gen) with precomputed input gates  (per step, after chain A).
#include "common.h"
#include <stdlib.h>

static inline int round4(int x) { return (x + 3) & ~3; }

static void lstm_reg(const DecoderArgs& a, SkinnyArgs& k, const uint8_t* hmask, const uint8_t* cmask, int t) {
    const long off = (long)t * a.B * a.H;
    if (a.zone) {
        if (a.training) { k.zone = 1; k.hmask = hmask ? hmask + off : nullptr; k.cmask = cmask ? cmask + off : nullptr; }
        else { k.zone = 2; k.zh = a.p_hidden; k.zc = a.p_cell; }
    } else if (a.training && hmask && a.p_hidden > 0.f) {
        k.zone = 0; k.hmask = hmask + off; k.hscale = 1.f / (1.f - a.p_hidden);
    }
}

// Operand descriptors: MFMA-tile-order ("packed") copies are used whenever the caller provided them.
static inline int bp16(int B) { return (B + 15) & ~15; }
static SkSeg seg_h(const float* rowmajor, const float* packed, int t, int B, int K, const float* w, const float* wp, int ldw) {
    SkSeg s; memset(&s, 0, sizeof(s));
    s.K = K;
    if (packed) { s.x = packed + (long)t * bp16(B) * K; s.xpack = 1; s.ldx = K; }
    else { s.x = rowmajor + (long)t * B * K; s.ldx = K; }
    if (wp) { s.w = wp; s.wpack = 1; s.ldw = K; } else { s.w = w; s.ldw = ldw; }
    return s;
}

// Chain B of the fast schedule for steps [c0, c1): generator-LSTM input gates (batched), the recurrent steps and the
// frame/stop projection (batched).  Runs on its own low-priority stream behind chain A (see side_stream()).
static bool gen_uses_lstep(const DecoderArgs& a) {
    return a.fast && a.gen_w2p && a.gen_bias_u && a.gen_w_ih_u && a.gate_part_gen && (a.H & 31) == 0;
}

// pre_gen[c0, c1) = [h_att, ctx] W_ih^T (batched over the chunk's steps)
static int gen_pre(const DecoderArgs& a, int c0, int c1, int region, hipStream_t s) {
    const int B = a.B, H = a.H, Dm = a.Dm, n = c1 - c0;
    const float* w_ih = gen_uses_lstep(a) ? a.gen_w_ih_u : a.gen_w_ih;      // unit-major rows -> unit-major pre_gen
    const long BH = (long)B * H, BD = (long)B * Dm, B4H = 4 * BH;
    GemmArgs g; memset(&g, 0, sizeof(g));
    g.taps = 1; g.batch = 1; g.zt = 1; g.alpha = 1.f; g.mask_scale = 1.f; g.nosplit = region;
    g.A = a.h_att + (c0 + 1) * BH; g.B = w_ih; g.C = a.pre_gen + c0 * B4H;
    g.M = n * B; g.N = 4 * H; g.K = H; g.Kc = H; g.lda = H; g.ldb = H + Dm; g.ldc = 4 * H; g.beta = 0.f;
    MTTS_TRY(mtts_gemm_ex(&g, s));
    g.A = a.ctx + (c0 + 1) * BD; g.B = w_ih + H; g.K = Dm; g.Kc = Dm; g.lda = Dm; g.beta = 1.f;
    return mtts_gemm_ex(&g, s);
}

// frame / stop projection of steps [c0, c1) (batched)
static int gen_proj(const DecoderArgs& a, int c0, int c1, int region, hipStream_t s) {
    const int B = a.B, M = a.M, H = a.H, Dm = a.Dm, Mo = round4(M + 1), n = c1 - c0;
    const long BH = (long)B * H, BD = (long)B * Dm;
    GemmArgs g; memset(&g, 0, sizeof(g));
    g.taps = 1; g.batch = 1; g.zt = 1; g.alpha = 1.f; g.mask_scale = 1.f; g.nosplit = region;
    g.A = a.h_gen + (c0 + 1) * BH; g.B = a.w_out; g.C = a.out + (long)(c0 + 1) * B * Mo; g.bias = a.b_out;
    g.M = n * B; g.N = M + 1; g.K = H; g.Kc = H; g.lda = H; g.ldb = H + Dm; g.ldc = Mo; g.beta = 0.f;
    MTTS_TRY(mtts_gemm_ex(&g, s));
    g.A = a.ctx + (c0 + 1) * BD; g.B = a.w_out + H; g.bias = nullptr; g.K = Dm; g.Kc = Dm; g.lda = Dm; g.beta = 1.f;
    return mtts_gemm_ex(&g, s);
}

// K-split step arguments of the generator LSTM (recurrent part; the input projection sits in pre_gen)
static LstmStepArgs gen_step_args(const DecoderArgs& a, int t) {
    const int B = a.B, H = a.H;
    const long BH = (long)B * H, B4H = 4 * BH;
    LstmStepArgs k; memset(&k, 0, sizeof(k));
    k.x[0] = a.h_gen + t * BH; k.K[0] = H; k.ldx[0] = H; k.nseg = 1;
    k.w_packed = a.gen_w2p; k.precision = ls_pack_mode(a.B, a.precision, !a.fast); k.B = B; k.H = H; k.partials = a.gate_part_gen; k.nb_max = 4;
    k.pre = a.pre_gen + t * B4H; k.ldpre = 4 * H; k.bias_u = a.gen_bias_u;
    k.h_prev = a.h_gen + t * BH; k.c_prev = a.c_gen + t * BH;
    k.h_out = a.h_gen + (t + 1) * BH; k.c_out = a.c_gen + (t + 1) * BH;
    k.gates_out = a.gates_gen ? a.gates_gen + t * B4H : nullptr;
    SkinnyArgs r; memset(&r, 0, sizeof(r));
    lstm_reg(a, r, a.gen_hmask, a.gen_cmask, t);
    k.hmask = r.hmask; k.cmask = r.cmask; k.hscale = r.hscale; k.zone = r.zone; k.zh = r.zh; k.zc = r.zc;
    return k;
}

// recurrent steps [c0, c1) of the generator LSTM on their own (side-stream chain / tail of the fused schedule)
static int gen_steps(const DecoderArgs& a, int c0, int c1, hipStream_t s) {
    const int B = a.B, H = a.H;
    const long BH = (long)B * H, B4H = 4 * BH;
    const bool use_ls = gen_uses_lstep(a);
    for (int t = c0; t < c1; ++t) {
        if (use_ls) { MTTS_TRY(lstm_step_launch(gen_step_args(a, t), s)); continue; }
        SkinnyArgs k; memset(&k, 0, sizeof(k));
        k.B = B; k.N = 4 * H; k.ksplit = 1; k.lstm = 1; k.H = H; k.nseg = 1;
        k.seg[0] = seg_h(a.h_gen, a.h_gen_p, t, B, H, a.gen_w_hh, a.gen_w_hh_p, H);
        k.pre = a.pre_gen + t * B4H; k.ldpre = 4 * H;
        k.b_ih = a.gen_b_ih; k.b_hh = a.gen_b_hh;
        k.h_prev = a.h_gen + t * BH; k.c_prev = a.c_gen + t * BH;
        k.h_out = a.h_gen + (t + 1) * BH; k.c_out = a.c_gen + (t + 1) * BH;
        k.gates_out = a.gates_gen ? a.gates_gen + t * B4H : nullptr;
        k.h_pack_out = a.h_gen_p ? a.h_gen_p + (long)(t + 1) * bp16(B) * H : nullptr;
        lstm_reg(a, k, a.gen_hmask, a.gen_cmask, t);
        MTTS_TRY(skinny_launch(k, s));
    }
    return 0;
}

// Chain B of the two-stream fast schedule for steps [c0, c1): input gates (batched), recurrent steps, projection (batched)
static int gen_chunk(const DecoderArgs& a, int c0, int c1, hipStream_t s) {
    MTTS_TRY(gen_pre(a, c0, c1, 1, s));
    MTTS_TRY(gen_steps(a, c0, c1, s));
    return gen_proj(a, c0, c1, 1, s);
}

MTTS_API int mtts_decoder_fwd(const DecoderArgs* args, void* stream) {
    const DecoderArgs& a = *args;
    hipStream_t s = (hipStream_t)stream;
    const int B = a.B, L = a.L, M = a.M, P = a.P, H = a.H, A = a.A, Dm = a.Dm;
    const int Mo = round4(M + 1);
    const long BH = (long)B * H, BD = (long)B * Dm, BL = (long)B * L, B4H = 4 * BH, BP = (long)B * P;
    MTTS_REQUIRE((P & 3) == 0 && (H & 3) == 0 && (Dm & 3) == 0 && (M & 3) == 0,
                 "decoder: P, H, Dm, num_mels must be multiples of 4 (P=%d H=%d Dm=%d M=%d)", P, H, Dm, M);
    MTTS_REQUIRE(a.n_prenet >= 1 && a.n_prenet <= 4, "decoder: 1..4 prenet layers supported");
    MTTS_REQUIRE(a.t0 >= 0 && a.t1 <= a.T && a.t0 <= a.t1, "decoder: bad step range");
    const float pscale = a.p_prenet > 0.f ? 1.f / (1.f - a.p_prenet) : 1.f;
    const int nsteps = a.t1 - a.t0;
    if (nsteps == 0) return 0;
    // attention LSTM as K-split gate GEMM + (cell, query partials): see lstm_step.hip
    const bool use_ls = a.fast && a.att_w2p && a.att_bias_u && a.att_w_pre_u && a.gate_part && (Dm & 31) == 0 && (H & 31) == 0 &&
                        (A & 15) == 0 && A <= 256;
    // general schedule (free running / mixed teacher forcing): the same kernels over the un-hoisted operands
    //   attention LSTM  [prenet(t) | ctx | h_att]  x  [W_ih | W_hh],   generator LSTM  [h_att | ctx | h_gen]  x  [W_ih | W_hh]
    const bool use_lg = !a.fast && a.att_w2p && a.att_bias_u && a.gate_part && a.gen_w2p && a.gen_bias_u && a.gate_part_gen &&
                        (P & 31) == 0 && (Dm & 31) == 0 && (H & 31) == 0 && (A & 15) == 0 && A <= 256;

    if (a.t0 == 0) {
        // U = W_loc [A,C] * W_conv [C,ksz];  Mt = memory W_memory^T;  PL[0] = Mt + bias   (attention.py:23-28)
        MTTS_TRY(gemm_plain(a.w_loc, a.w_conv, a.U, A, a.ksz, a.C, a.C, a.ksz, a.ksz, false, true, 1.f, 0.f, nullptr, 0, s));
        MTTS_TRY(gemm_plain(a.memory, a.w_memory, a.Mt, B * L, A, Dm, Dm, Dm, A, false, false, 1.f, 0.f, nullptr, 0, s));
        MTTS_TRY(attn_pl_init(a.Mt, a.att_bias, a.PL, BL * A, A, s));
        if (a.att_w_ctx_p) MTTS_TRY(mtts_pack_weight(a.att_w_ih + P, P + Dm, 4 * H, Dm, H, a.att_w_ctx_p, s));
        if (a.att_w_hh_p) MTTS_TRY(mtts_pack_weight(a.att_w_hh, H, 4 * H, H, H, a.att_w_hh_p, s));
        if (a.gen_w_hh_p) MTTS_TRY(mtts_pack_weight(a.gen_w_hh, H, 4 * H, H, H, a.gen_w_hh_p, s));
        if (a.w_query_p) MTTS_TRY(mtts_pack_weight(a.w_query, H, A, H, 0, a.w_query_p, s));
        if (use_ls) {     // K-split step path: packed [W_ih[:, P:] | W_hh], unit-major bias and hoisted-projection rows
            LstmPackArgs k; memset(&k, 0, sizeof(k));
            k.w[0] = a.att_w_ih + P; k.K[0] = Dm; k.ldw[0] = P + Dm;
            k.w[1] = a.att_w_hh; k.K[1] = H; k.ldw[1] = H;
            k.nseg = 2; k.H = H; k.precision = ls_pack_mode(a.B, a.precision, !a.fast); k.dst = a.att_w2p;
            k.b_ih = a.att_b_ih; k.b_hh = a.att_b_hh; k.bias_u = a.att_bias_u;
            MTTS_TRY(mtts_lstm_pack_weights(&k, s));
            MTTS_TRY(mtts_lstm_rows_unit_major(a.att_w_ih, P + Dm, H, P, a.att_w_pre_u, s));
        }
        if (a.n_prenet == 2 && a.prenet_wp[0] && a.prenet_wp[1] && (M & 15) == 0 && (P & 15) == 0) {
            MTTS_TRY(mtts_pack_weight(a.prenet_w[0], M, P, M, 0, a.prenet_wp[0], s));
            MTTS_TRY(mtts_pack_weight(a.prenet_w[1], P, P, P, 0, a.prenet_wp[1], s));
        }
        if (use_lg) {
            LstmPackArgs k; memset(&k, 0, sizeof(k));
            k.w[0] = a.att_w_ih; k.K[0] = P; k.ldw[0] = P + Dm;
            k.w[1] = a.att_w_ih + P; k.K[1] = Dm; k.ldw[1] = P + Dm;
            k.w[2] = a.att_w_hh; k.K[2] = H; k.ldw[2] = H;
            k.nseg = 3; k.H = H; k.precision = ls_pack_mode(a.B, a.precision, !a.fast); k.dst = a.att_w2p;
            k.b_ih = a.att_b_ih; k.b_hh = a.att_b_hh; k.bias_u = a.att_bias_u;
            MTTS_TRY(mtts_lstm_pack_weights(&k, s));
            k.w[0] = a.gen_w_ih; k.K[0] = H; k.ldw[0] = H + Dm;
            k.w[1] = a.gen_w_ih + H; k.K[1] = Dm; k.ldw[1] = H + Dm;
            k.w[2] = a.gen_w_hh; k.K[2] = H; k.ldw[2] = H;
            k.dst = a.gen_w2p; k.b_ih = a.gen_b_ih; k.b_hh = a.gen_b_hh; k.bias_u = a.gen_bias_u;
            MTTS_TRY(mtts_lstm_pack_weights(&k, s));
        }
        if (gen_uses_lstep(a)) {
            LstmPackArgs k; memset(&k, 0, sizeof(k));
            k.w[0] = a.gen_w_hh; k.K[0] = H; k.ldw[0] = H; k.nseg = 1; k.H = H; k.precision = ls_pack_mode(a.B, a.precision, !a.fast); k.dst = a.gen_w2p;
            k.b_ih = a.gen_b_ih; k.b_hh = a.gen_b_hh; k.bias_u = a.gen_bias_u;
            MTTS_TRY(mtts_lstm_pack_weights(&k, s));
            MTTS_TRY(mtts_lstm_rows_unit_major(a.gen_w_ih, H + Dm, H, H + Dm, a.gen_w_ih_u, s));
        }
    }

    // prenet over the teacher frames of this range (tacotron2.py:126-133)
    if (a.frames_in) {
        for (int i = 0; i < a.n_prenet; ++i) {
            GemmArgs g; memset(&g, 0, sizeof(g));
            const int Kin = i == 0 ? M : P;
            g.A = (i == 0 ? a.frames_in + (long)a.t0 * B * M : a.prenet_act[i - 1] + a.t0 * BP);
            g.B = a.prenet_w[i]; g.C = a.prenet_act[i] + a.t0 * BP; g.bias = a.prenet_b[i];
            g.M = nsteps * B; g.N = P; g.K = Kin; g.lda = Kin; g.ldb = Kin; g.ldc = P;
            g.taps = 1; g.Kc = Kin; g.batch = 1; g.zt = 1; g.alpha = 1.f; g.act = MTTS_ACT_RELU;
            if (a.prenet_mask[i] && a.p_prenet > 0.f) { g.mask = a.prenet_mask[i] + a.t0 * BP; g.ldmask = P; g.mask_scale = pscale; }
            MTTS_TRY(mtts_gemm_ex(&g, s));
        }
    }
    float* pren = a.prenet_act[a.n_prenet - 1];

    if (a.fast) {
        MTTS_REQUIRE(a.pre_att && a.pre_gen && a.frames_in, "decoder fast path needs pre_att/pre_gen workspaces and frames_in");
        MTTS_TRY(gemm_plain(pren + a.t0 * BP, use_ls ? a.att_w_pre_u : a.att_w_ih, a.pre_att + a.t0 * B4H, nsteps * B, 4 * H, P, P,
                            use_ls ? P : P + Dm, 4 * H, false, false, 1.f, 0.f, nullptr, 0, s));
    }

    // fast schedule: chain B trails chain A by one chunk on the side stream (sharing launches between the two chains was measured
    // slower: the side stream fills exactly the ramp / drain / boundary gaps of chain A)
    const int CH = decoder_chunk();
    // persistent generator LSTM (persist.hip): chain B runs AFTER chain A as input GEMM (all steps) -> one persistent launch
    // -> projection GEMM (all steps), everything on the caller's stream
    const bool pg = a.fast && gen_uses_lstep(a) && pgen_supported(a);
    hipStream_t sb = (a.fast && !pg) ? side_stream(s) : nullptr;
    if (a.fast && !pg && !sb) return mtts_fail("decoder: cannot create the side stream");
    // persistent attention LSTM + attention (persist.hip): chain A of the fast schedule as ONE launch
    const bool pd = pg && use_ls && pdec_supported(a);
    if (pd) {      // (bench.py samples the whole launch with HIP events: mtts_prof_begin / mtts_prof_end)
        const bool sampled = prof_sample(0, s, 0);
        MTTS_TRY(pdec_launch(a, a.t0, a.t1, s));
        if (sampled) prof_sample(0, s, 1);
    }
    for (int t = a.t0; t < a.t1 && !pd; ++t) {
        const bool teach = a.frames_in && a.teacher && a.teacher[t];
        if (!teach) {
            // prenet on the model's own previous frame (tacotron2.py:181), out slot t holds frame t-1 (slot 0 = zeros)
            int fused = -1;
            if (a.n_prenet == 2 && a.prenet_wp[0] && a.prenet_wp[1] && (M & 15) == 0 && (P & 15) == 0) {      // both layers in one launch
                const bool masked = a.p_prenet > 0.f;
                fused = prenet2_launch(a.out + (long)t * B * Mo, Mo, M, a.prenet_wp[0], a.prenet_b[0], a.prenet_wp[1], a.prenet_b[1],
                                       masked && a.prenet_mask[0] ? a.prenet_mask[0] + t * BP : nullptr,
                                       masked && a.prenet_mask[1] ? a.prenet_mask[1] + t * BP : nullptr, pscale,
                                       a.prenet_act[0] + t * BP, a.prenet_act[1] + t * BP, B, P, s);
                if (fused > 0) return fused;
            }
            for (int i = 0; fused < 0 && i < a.n_prenet; ++i) {
                SkinnyArgs k; memset(&k, 0, sizeof(k));
                k.nseg = 1; k.B = B; k.N = P; k.ksplit = 1;
                if (i == 0) k.seg[0] = SkSeg{a.out + (long)t * B * Mo, a.prenet_w[0], M, Mo, M, 0, 0};
                else k.seg[0] = SkSeg{a.prenet_act[i - 1] + t * BP, a.prenet_w[i], P, P, P, 0, 0};
                k.out = a.prenet_act[i] + t * BP; k.ldo = P; k.bias = a.prenet_b[i]; k.act = MTTS_ACT_RELU;
                if (a.prenet_mask[i] && a.p_prenet > 0.f) { k.mask = a.prenet_mask[i] + t * BP; k.ldmask = P; k.mask_scale = pscale; }
                MTTS_TRY(skinny_launch(k, s));
            }
        }
        if (use_ls || use_lg) {   // attention LSTM + query partials (tacotron2.py:184-185, attention.py:68) in two launches
            LstmStepArgs k; memset(&k, 0, sizeof(k));
            int n = 0;
            if (use_lg) { k.x[n] = pren + t * BP; k.K[n] = P; k.ldx[n] = P; ++n; }
            k.x[n] = a.ctx + t * BD; k.K[n] = Dm; k.ldx[n] = Dm; ++n;
            k.x[n] = a.h_att + t * BH; k.K[n] = H; k.ldx[n] = H; ++n;
            k.nseg = n; k.w_packed = a.att_w2p; k.precision = ls_pack_mode(a.B, a.precision, !a.fast); k.B = B; k.H = H; k.partials = a.gate_part;
            if (use_ls) { k.pre = a.pre_att + t * B4H; k.ldpre = 4 * H; k.nb_max = 4; }      // training: short slices, 2 workgroups per CU
            k.bias_u = a.att_bias_u;
            k.h_prev = a.h_att + t * BH; k.c_prev = a.c_att + t * BH;
            k.h_out = a.h_att + (t + 1) * BH; k.c_out = a.c_att + (t + 1) * BH;
            k.gates_out = a.gates_att ? a.gates_att + t * B4H : nullptr;
            {
                SkinnyArgs r; memset(&r, 0, sizeof(r));
                lstm_reg(a, r, a.att_hmask, a.att_cmask, t);
                k.hmask = r.hmask; k.cmask = r.cmask; k.hscale = r.hscale; k.zone = r.zone; k.zh = r.zh; k.zc = r.zc;
            }
            k.w_query = a.w_query; k.A = A; k.qpart = a.qpart;
            const bool sampled = prof_sample(t, s, 0);
            MTTS_TRY(lstm_step_launch(k, s));
            if (sampled) prof_sample(t, s, 1);
        } else {   // attention LSTM (tacotron2.py:184-185)
            SkinnyArgs k; memset(&k, 0, sizeof(k));
            k.B = B; k.N = 4 * H; k.ksplit = 1; k.lstm = 1; k.H = H;
            if (a.fast) {
                k.nseg = 2;
                k.seg[0] = seg_h(a.ctx, a.ctx_p, t, B, Dm, a.att_w_ih + P, a.att_w_ctx_p, P + Dm);
                k.seg[1] = seg_h(a.h_att, a.h_att_p, t, B, H, a.att_w_hh, a.att_w_hh_p, H);
                k.pre = a.pre_att + t * B4H; k.ldpre = 4 * H;
            } else {
                k.nseg = 3;
                k.seg[0] = SkSeg{pren + t * BP, a.att_w_ih, P, P, P + Dm, 0, 0};
                k.seg[1] = seg_h(a.ctx, a.ctx_p, t, B, Dm, a.att_w_ih + P, a.att_w_ctx_p, P + Dm);
                k.seg[2] = seg_h(a.h_att, a.h_att_p, t, B, H, a.att_w_hh, a.att_w_hh_p, H);
            }
            k.b_ih = a.att_b_ih; k.b_hh = a.att_b_hh;
            k.h_prev = a.h_att + t * BH; k.c_prev = a.c_att + t * BH;
            k.h_out = a.h_att + (t + 1) * BH; k.c_out = a.c_att + (t + 1) * BH;
            k.gates_out = a.gates_att ? a.gates_att + t * B4H : nullptr;
            k.h_pack_out = a.h_att_p ? a.h_att_p + (long)(t + 1) * bp16(B) * H : nullptr;
            lstm_reg(a, k, a.att_hmask, a.att_cmask, t);
            const bool sampled = prof_sample(t, s, 0);
            MTTS_TRY(skinny_launch(k, s));
            if (sampled) prof_sample(t, s, 1);
        }
        if (!use_ls && !use_lg) {   // query projection partials (attention.py:68)
            SkinnyArgs k; memset(&k, 0, sizeof(k));
            k.nseg = 1; k.B = B; k.N = A; k.ksplit = a.kq;
            k.seg[0] = seg_h(a.h_att, a.h_att_p, t + 1, B, H, a.w_query, a.w_query_p, H);
            k.out = a.qpart; k.ldo = A; k.out_ks = (long)B * A;
            MTTS_TRY(skinny_launch(k, s));
        }
        {   // energies -> softmax -> context, PL for the next step (attention.py:39-45,67-86)
            AttnStepArgs q; memset(&q, 0, sizeof(q));
            q.qpart = a.qpart; q.kq = (use_ls || use_lg) ? H / 16 : a.kq; q.q_ks = (long)B * A;
            q.PL = a.PL + (long)(t & 1) * BL * A; q.PL_next = a.PL + (long)((t + 1) & 1) * BL * A;
            q.Mt = a.Mt; q.U = a.U; q.bias = a.att_bias; q.v = a.w_energy; q.memory = a.memory; q.lengths = a.lengths;
            q.cum_in = a.cum + t * BL; q.cum_out = a.cum + (t + 1) * BL; q.w_out = a.align + t * BL;
            q.ctx_out = a.ctx + (t + 1) * BD;
            q.q_out = a.q_all ? a.q_all + (long)t * B * A : nullptr;
            q.ctx_pack_out = a.ctx_p ? a.ctx_p + (long)(t + 1) * bp16(B) * Dm : nullptr;
            q.B = B; q.L = L; q.A = A; q.Dm = Dm; q.ksz = a.ksz;
            q.nch = attn_step_nch(B, L, A, Dm, a.ksz, q.kq);
            MTTS_TRY(attn_step_launch(q, s));
        }
        if (!a.fast) {
            if (use_lg) {   // generator LSTM (tacotron2.py:187-188)
                LstmStepArgs k; memset(&k, 0, sizeof(k));
                k.x[0] = a.h_att + (t + 1) * BH; k.K[0] = H; k.ldx[0] = H;
                k.x[1] = a.ctx + (t + 1) * BD; k.K[1] = Dm; k.ldx[1] = Dm;
                k.x[2] = a.h_gen + t * BH; k.K[2] = H; k.ldx[2] = H;
                k.nseg = 3; k.w_packed = a.gen_w2p; k.precision = ls_pack_mode(a.B, a.precision, !a.fast); k.B = B; k.H = H; k.partials = a.gate_part_gen;
                k.bias_u = a.gen_bias_u;
                k.h_prev = a.h_gen + t * BH; k.c_prev = a.c_gen + t * BH;
                k.h_out = a.h_gen + (t + 1) * BH; k.c_out = a.c_gen + (t + 1) * BH;
                k.gates_out = a.gates_gen ? a.gates_gen + t * B4H : nullptr;
                SkinnyArgs r; memset(&r, 0, sizeof(r));
                lstm_reg(a, r, a.gen_hmask, a.gen_cmask, t);
                k.hmask = r.hmask; k.cmask = r.cmask; k.hscale = r.hscale; k.zone = r.zone; k.zh = r.zh; k.zc = r.zc;
                MTTS_TRY(lstm_step_launch(k, s));
            } else {   // generator LSTM (tacotron2.py:187-188)
                SkinnyArgs k; memset(&k, 0, sizeof(k));
                k.B = B; k.N = 4 * H; k.ksplit = 1; k.lstm = 1; k.H = H; k.nseg = 3;
                k.seg[0] = seg_h(a.h_att, a.h_att_p, t + 1, B, H, a.gen_w_ih, nullptr, H + Dm);
                k.seg[1] = seg_h(a.ctx, a.ctx_p, t + 1, B, Dm, a.gen_w_ih + H, nullptr, H + Dm);
                k.seg[2] = seg_h(a.h_gen, a.h_gen_p, t, B, H, a.gen_w_hh, a.gen_w_hh_p, H);
                k.b_ih = a.gen_b_ih; k.b_hh = a.gen_b_hh;
                k.h_prev = a.h_gen + t * BH; k.c_prev = a.c_gen + t * BH;
                k.h_out = a.h_gen + (t + 1) * BH; k.c_out = a.c_gen + (t + 1) * BH;
                k.gates_out = a.gates_gen ? a.gates_gen + t * B4H : nullptr;
                k.h_pack_out = a.h_gen_p ? a.h_gen_p + (long)(t + 1) * bp16(B) * H : nullptr;
                lstm_reg(a, k, a.gen_hmask, a.gen_cmask, t);
                MTTS_TRY(skinny_launch(k, s));
            }
            {   // frame + stop projection (tacotron2.py:191-193)
                SkinnyArgs k; memset(&k, 0, sizeof(k));
                k.nseg = 2; k.B = B; k.N = M + 1; k.ksplit = 1;
                k.seg[0] = seg_h(a.h_gen, use_lg ? nullptr : a.h_gen_p, t + 1, B, H, a.w_out, nullptr, H + Dm);      // (the K-split cell kernel
                k.seg[1] = seg_h(a.ctx, use_lg ? nullptr : a.ctx_p, t + 1, B, Dm, a.w_out + H, nullptr, H + Dm);      //  writes no packed copies)
                // few output columns (6 tiles), long K: 16-row workgroups with every K chunk in flight (skinny_kernel_wide) - one
                // launch and one memory round trip (round 4; before: an 8-way K split + a slab-sum launch, 6.8 + 5.4 us at batch 128)
                k.out = a.out + (long)(t + 1) * B * Mo; k.ldo = Mo; k.bias = a.b_out;
                MTTS_TRY(skinny_launch(k, s));
            }
        }
        if (a.fast && !pg && (((t + 1 - a.t0) % CH) == 0 || t + 1 == a.t1)) {
            const int c1 = t + 1, c0 = a.t0 + ((c1 - a.t0 - 1) / CH) * CH;
            hipEvent_t ev = pool_event(s);
            MTTS_CHECK_HIP(hipEventRecord(ev, s));
            MTTS_CHECK_HIP(hipStreamWaitEvent(sb, ev, 0));
            MTTS_TRY(gen_chunk(a, c0, c1, sb));
        }
    }
    if (pg) {
        MTTS_TRY(gen_pre(a, a.t0, a.t1, 0, s));
        MTTS_TRY(pgen_launch(a, a.t0, a.t1, s));
        return gen_proj(a, a.t0, a.t1, 0, s);
    }
    if (a.fast) {     // join: the caller's stream continues after chain B
        hipEvent_t ev = pool_event(s);
        MTTS_CHECK_HIP(hipEventRecord(ev, sb));
        MTTS_CHECK_HIP(hipStreamWaitEvent(s, ev, 0));
        return 0;
    }

    return 0;
}

MTTS_API int mtts_bilstm_fwd(const BiLstmArgs* args, void* stream) {
    // Time-major everywhere: x [L,B,Cin], xproj/gates [L,B,4H] indexed by the time step t.
    // State arrays [L+1,B,H]: forward direction reads slot t / writes slot t+1; the reverse direction walks
    // t = L-1..0, reads slot t+1 / writes slot t (slot L = zero initial state).
    const BiLstmArgs& a = *args;
    hipStream_t s = (hipStream_t)stream;
    const int B = a.B, L = a.L, H = a.H;
    MTTS_REQUIRE((H & 3) == 0 && (a.Cin & 3) == 0, "bilstm: H and Cin must be multiples of 4");
    const long BH = (long)B * H;
    // the two directions are independent (disjoint state, gate and output-column arrays): the reverse one runs on the side stream
    hipStream_t sd[2] = {s, side_stream(s)};
    if (!sd[1]) return mtts_fail("bilstm: cannot create the side stream");
    hipEvent_t ev_fork = pool_event(s), ev_join = pool_event(s);
    MTTS_CHECK_HIP(hipEventRecord(ev_fork, s));
    MTTS_CHECK_HIP(hipStreamWaitEvent(sd[1], ev_fork, 0));
    for (int d = 1; d >= 0; --d) {
        {   // input projection for all steps (K = Cin is short: never split-K, so no shared scratch between the streams)
            GemmArgs q; memset(&q, 0, sizeof(q));
            q.A = a.x; q.B = a.w_ih[d]; q.C = a.xproj[d]; q.M = B * L; q.N = 4 * H; q.K = a.Cin; q.Kc = a.Cin;
            q.lda = a.Cin; q.ldb = a.Cin; q.ldc = 4 * H; q.taps = 1; q.batch = 1; q.zt = 1; q.alpha = 1.f; q.mask_scale = 1.f; q.nosplit = d;
            MTTS_TRY(mtts_gemm_ex(&q, sd[d]));
        }
        for (int st = 0; st < L; ++st) {
            const int t = d == 0 ? st : L - 1 - st;
            const int s_in = d == 0 ? t : t + 1, s_out = d == 0 ? t + 1 : t;
            SkinnyArgs k; memset(&k, 0, sizeof(k));
            k.B = B; k.N = 4 * H; k.ksplit = 1; k.lstm = 1; k.H = H; k.nseg = 1;
            k.seg[0] = SkSeg{a.h[d] + s_in * BH, a.w_hh[d], H, H, H, 0, 0};
            k.pre = a.xproj[d] + (long)t * 4 * BH; k.ldpre = 4 * H;
            k.b_ih = a.b_ih[d]; k.b_hh = a.b_hh[d];
            k.h_prev = a.h[d] + s_in * BH; k.c_prev = a.c[d] + s_in * BH;
            k.h_out = a.h[d] + s_out * BH; k.c_out = a.c[d] + s_out * BH;
            k.gates_out = a.gates[d] ? a.gates[d] + (long)t * 4 * BH : nullptr;
            k.lengths = a.lengths; k.t = t;
            k.y_out = a.y + (long)t * 2 * H + d * H; k.ldy = L * 2 * H;
            MTTS_TRY(skinny_launch(k, sd[d]));
        }
    }
    MTTS_CHECK_HIP(hipEventRecord(ev_join, sd[1]));
    MTTS_CHECK_HIP(hipStreamWaitEvent(s, ev_join, 0));
    return 0;
}
