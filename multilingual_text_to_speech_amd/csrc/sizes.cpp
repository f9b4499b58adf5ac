// Buffer-size queries of the C-ABI (host only): how many ELEMENTS the caller must allocate for every caller-provided buffer of
// DecoderArgs / DecoderGradArgs / BiLstmArgs / BiLstmGradArgs, by field name.  A binding fills the shape fields of the argument
// block (B, L, T, M, P, H, A, Dm, ksz, C, n_prenet, kq, fast, precision; ksb / ksb_ctx / nch of the gradient block) and asks here
// instead of re-deriving the formulas of the header comments.  Element = float unless the field is documented in bytes
// (att_w2p / gen_w2p / persist_ws: bytes; masks: uint8 flags).  Returns -1 for an unknown field name.
#include "common.h"

static inline long r4(long x) { return (x + 3) & ~3L; }
static inline long r16(long x) { return (x + 15) & ~15L; }

// K of the packed step-kernel weights / partial slabs for the schedule the argument block selects
static long att_k(const DecoderArgs& a) { return a.fast ? a.Dm + a.H : a.P + a.Dm + a.H; }
static long gen_k(const DecoderArgs& a) { return a.fast ? a.H : 2L * a.H + a.Dm; }

MTTS_API long mtts_decoder_buffer_elems(const DecoderArgs* args, const char* field) {
    const DecoderArgs& a = *args;
    const long B = a.B, L = a.L, T = a.T, M = a.M, P = a.P, H = a.H, A = a.A, Dm = a.Dm, Mo = r4(M + 1), Bp = r16(B);
    struct Row { const char* name; long n; };
    const Row rows[] = {
        {"prenet_act", T * B * P}, {"U", A * (long)a.ksz}, {"Mt", B * L * A}, {"PL", 2 * B * L * A},
        {"qpart", (long)(a.kq > H / 16 ? a.kq : H / 16) * B * A},
        {"h_att", (T + 1) * B * H}, {"c_att", (T + 1) * B * H}, {"h_gen", (T + 1) * B * H}, {"c_gen", (T + 1) * B * H},
        {"ctx", (T + 1) * B * Dm}, {"cum", (T + 1) * B * L}, {"align", T * B * L},
        {"gates_att", T * B * 4 * H}, {"gates_gen", T * B * 4 * H}, {"out", (T + 1) * B * Mo},
        {"pre_att", T * B * 4 * H}, {"pre_gen", T * B * 4 * H}, {"q_all", T * B * A},
        {"h_att_p", (T + 1) * Bp * H}, {"h_gen_p", (T + 1) * Bp * H}, {"ctx_p", (T + 1) * Bp * Dm},
        {"att_w_ctx_p", 4 * H * Dm}, {"att_w_hh_p", 4 * H * H}, {"gen_w_hh_p", 4 * H * H}, {"w_query_p", r16(A) * H},
        {"att_w2p", mtts_lstm_packed_weight_bytes((int)H, (int)att_k(a), ls_pack_mode((int)B, a.precision, !a.fast))}, {"att_bias_u", 4 * H}, {"att_w_pre_u", 4 * H * P},
        {"gate_part", mtts_lstm_step_partial_floats((int)B, (int)H, (int)att_k(a))},
        {"gen_w2p", mtts_lstm_packed_weight_bytes((int)H, (int)gen_k(a), ls_pack_mode((int)B, a.precision, !a.fast))}, {"gen_bias_u", 4 * H}, {"gen_w_ih_u", 4 * H * (H + Dm)},
        {"gate_part_gen", mtts_lstm_step_partial_floats((int)B, (int)H, (int)gen_k(a))},
        {"prenet_wp0", P * M}, {"prenet_wp1", P * P},
        {"persist_ws", mtts_decoder_persist_ws_bytes((int)B, (int)L, (int)H, (int)Dm, (int)A)},
        {"prenet_mask", T * B * P}, {"att_hmask", T * B * H}, {"att_cmask", T * B * H}, {"gen_hmask", T * B * H}, {"gen_cmask", T * B * H},
        {"frames_in", T * B * M},
    };
    for (const Row& r : rows)
        if (strcmp(r.name, field) == 0) return r.n;
    return -1;
}

MTTS_API long mtts_decoder_grad_buffer_elems(const DecoderArgs* fwd, const DecoderGradArgs* grad, const char* field) {
    const DecoderArgs& a = *fwd;
    const DecoderGradArgs& g = *grad;
    const long B = a.B, L = a.L, T = a.T, M = a.M, P = a.P, H = a.H, A = a.A, Dm = a.Dm, Mo = r4(M + 1), Bp = r16(B), n = a.n_prenet;
    const long ksb = g.ksb, ksc = g.ksb_ctx > 0 ? g.ksb_ctx : g.ksb, nch = g.nch;
    long cmax = 4 * H; if (A * a.ksz > cmax) cmax = A * (long)a.ksz; if (P > cmax) cmax = P; if (M + 1 > cmax) cmax = M + 1;
    struct Row { const char* name; long n; };
    const Row rows[] = {
        {"dout", (T + 1) * B * Mo}, {"dalign", T * B * L},
        {"att_w_rec_T", (Dm + H) * 4 * H}, {"att_w_ih_T", (P + Dm + H) * 4 * H}, {"gen_w_hh_T", H * 4 * H}, {"gen_w_ih_T", (2 * H + Dm) * 4 * H},
        {"w_out_T", (H + Dm) * Mo}, {"prenet_w_T0", M * P}, {"prenet_w_T", P * P},
        {"step_ws", B * ((P + Dm + H) + (2 * H + Dm) + (H + Dm) + M)}, {"frames_fed", T * B * M}, {"w_query_T", H * A},
        {"dG_att", T * B * 4 * H}, {"dG_gen", T * B * 4 * H}, {"dG_att_p", T * Bp * 4 * H}, {"dG_gen_p", T * Bp * 4 * H},
        {"att_w_rec_Tp", r16(Dm + H) * 4 * H}, {"gen_w_hh_Tp", H * 4 * H},
        {"dHG", T * B * H}, {"dHA", T * B * H}, {"dctx_all", (T + 1) * B * Dm}, {"dctx_tot", (T + 1) * B * Dm}, {"dcum_all", (T + 1) * B * L},
        {"dq_all", T * B * A}, {"part_gen", ksb * B * H}, {"part_att", ksc * B * Dm + ksb * B * H},
        {"dc_att", 2 * B * H}, {"dc_gen", 2 * B * H}, {"dh_carry_att", 2 * B * H}, {"dh_carry_gen", 2 * B * H},
        {"dMt", B * L * A}, {"dU_slab", B * nch * A * a.ksz}, {"dv_slab", B * nch * A}, {"dbias_slab", B * nch * A}, {"dU", A * (long)a.ksz},
        {"dpren", n * T * B * P}, {"colsum_ws", mtts_colsum_workspace_floats((int)cmax)}, {"dmemory", B * L * Dm},
        {"part_ring", (long)g.part_ring_slots * (ksc * B * Dm + ksb * B * H)},
    };
    for (const Row& r : rows)
        if (strcmp(r.name, field) == 0) return r.n;
    return -1;
}

// BiLstmArgs / BiLstmGradArgs buffers (ksb = K-splits of the recurrent input-gradient product, BiLstmGradArgs.ksb)
MTTS_API long mtts_bilstm_buffer_elems(const BiLstmArgs* args, int ksb, const char* field) {
    const BiLstmArgs& a = *args;
    const long B = a.B, L = a.L, H = a.H, Cin = a.Cin;
    struct Row { const char* name; long n; };
    const Row rows[] = {
        {"x", L * B * Cin}, {"xproj", L * B * 4 * H}, {"h", (L + 1) * B * H}, {"c", (L + 1) * B * H}, {"gates", L * B * 4 * H}, {"y", B * L * 2 * H},
        {"w_hh_T", H * 4 * H}, {"dxproj", L * B * 4 * H}, {"part", 2L * ksb * B * H}, {"dc", 2L * 2 * B * H}, {"dh_carry", 2L * 2 * B * H},
        {"colsum_ws", mtts_colsum_workspace_floats((int)(4 * H))}, {"dx", L * B * Cin}, {"dy", B * L * 2 * H},
    };
    for (const Row& r : rows)
        if (strcmp(r.name, field) == 0) return r.n;
    return -1;
}
