/* libmtts_hip -- C ABI of the MI355X-native multilingual Tacotron-2 text->mel hot path.
 *
 * The reference (Tomiinek/Multilingual_Text_to_Speech) has no FFI: its hot path is a chain of ATen ops
 * behind torch.nn.Module.  Each entry point below names the reference call sites it replaces.
 * Conventions: raw device pointers, explicit sizes/strides in ELEMENTS, `stream` is a hipStream_t;
 * return 0 on success, non-zero on error (text via mtts_last_error(), thread-local); functions never
 * allocate, never synchronise the stream and are re-entrant per (device, stream).  fp32 throughout.
 * Activations are channel-last: a conv input [N, C, L] of the reference is stored [N*L, C].
 * Dropout masks are inputs (uint8 keep flags + a scale) so that tests can inject the reference's draws.
 *
 * This header is also parsed by multilingual_text_to_speech_amd/_C.py to build the ctypes mirrors of
 * the structs: keep every field declaration of the form `<type> <name>;` with the types used below.
 */
#ifndef MTTS_H
#define MTTS_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { MTTS_ACT_NONE = 0, MTTS_ACT_RELU = 1, MTTS_ACT_TANH = 2, MTTS_ACT_SIGMOID = 3 };

/* ---- general fp32 MFMA GEMM:  C[M,N] = epilogue(alpha * sum_k A(m,k) B(n,k)) ------------------------------
 * Replaces torch.nn.Linear / F.conv1d / their autograd GEMMs at: modules/tacotron2.py:34,111-112 (prenet,
 * frame/stop projection), modules/attention.py:18-20,63 (memory/location projections), modules/layers.py:75
 * (Conv1d inside ConvBlock), modules/generated.py:42 (F.conv1d with generated kernels), modules/encoder.py:33
 * (LSTM input projection), modules/classifier.py:53-54.
 * transX = 0: operand stored [rows][K] (K contiguous); 1: stored [K][rows].
 * Convolution = implicit GEMM over channel-last rows r = n*seq_len + l: K = taps*Kc, tap t reads row
 * r + shift0 + t*dshift and contributes zero outside [0, seq_len) ('same' zero padding, layers.py:72-74). */
typedef struct GemmArgs {
    const float* A;
    const float* B;
    float* C;
    const float* bias;
    const uint8_t* mask;
    int M;
    int N;
    int K;
    int lda;
    int ldb;
    int ldc;
    int ldmask;
    int transA;
    int transB;
    int taps;
    int Kc;
    int seq_len;
    int shift_mode;   /* 0 none; 1: A rows shifted per tap (conv fwd / bwd-data); 2: B k-rows shifted by the z tap (wgrad) */
    int shift0;
    int dshift;
    long b_tap;       /* B element offset per tap (transB conv bwd-data) */
    int batch;        /* grid.z = batch * zt */
    int zt;           /* >1: z % zt selects the tap (shift, C offset c_ztap) for weight gradients */
    long a_z;
    long b_z;
    long c_z;
    long bias_z;
    long c_ztap;
    float alpha;
    float beta;
    int act;
    float mask_scale;
    int nosplit;      /* scratch region for split-K partial tiles: 0 = caller's stream, 1 = the library's side stream, 2 = its
                         weight-gradient stream (each third of the arena belongs to one of them; the name is historical) */
    int precision;    /* 0: the process-wide default set by mtts_set_precision (fp32 unless changed; what library-internal GEMMs
                         use); 1: bf16 path - operands rounded to bf16 (RNE) when their tile is staged, ONE MFMA term, fp32
                         accumulation and output; 2: fp32-accurate (six bf16 MFMA terms of exact 3-way operand splits) */
} GemmArgs;

int mtts_gemm_ex(const GemmArgs* args, void* stream);
/* Process-wide default precision (0 = fp32-accurate, 1 = bf16 operands) of the library-internal contractions: the batched GEMMs inside
 * mtts_decoder_fwd/bwd and mtts_bilstm_fwd/bwd.  The per-step kernels take theirs from DecoderArgs.precision. */
int mtts_set_precision(int precision);
int mtts_get_precision(void);
/* Scratch arena for split-K partial tiles, provided by the caller; NULL disables split-K.  (The library allocates device memory in ONE
 * place only: the pack buffer of the pre-split GEMM core, bf16 mode - see mtts_set_planes_workspace below.)
 * mtts_set_workspace binds the arena to the CURRENT device (default for all of its streams); mtts_set_stream_workspace
 * gives one caller stream its own arena, which is what makes concurrent callers on different streams of one device
 * independent: helper streams, ordering events and scratch are all looked up by (device, caller stream). */
int mtts_set_workspace(void* ptr, size_t bytes);
int mtts_set_stream_workspace(void* stream, void* ptr, size_t bytes);
/* Pack buffer of the pre-split GEMM core (bf16 mode's large plain GEMMs; csrc/gemm_planes.h): per (device, stream).  By default the
 * library hipMallocs it on first use and grows it in 64 MB granules (hundreds of MB per stream at 38400 x 4096 operands, invisible to
 * the caller's allocator).  mtts_set_planes_workspace hands the stream a CALLER-OWNED arena instead (never resized: a GEMM whose pack
 * does not fit runs on the core that needs no pack pass); ptr = NULL returns the stream to the library-owned buffer.  mtts_planes_trim
 * frees every library-owned pack buffer of the current device (synchronises their streams) and returns the bytes released. */
int mtts_set_planes_workspace(void* stream, void* ptr, size_t bytes);
size_t mtts_planes_trim(void);
int mtts_gemm(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
              int transA, int transB, float alpha, float beta, const float* bias, int act, void* stream);

/* ---- BatchNorm + activation + dropout (+ highway gate), channel-last, fwd/bwd ------------------------------
 * Replaces BatchNorm1d/F.batch_norm + activation + Dropout of modules/layers.py:78-86, modules/generated.py:94-96
 * and the highway combination modules/layers.py:149-153,174-178.  Statistics cover all R rows (padding included). */
typedef struct BnArgs {
    const float* x;        /* [R, C] conv output */
    const float* gamma;
    const float* beta;
    float* running_mean;   /* updated when training (nullable in bwd) */
    float* running_var;
    float* save_mean;      /* [C] written by fwd, read by bwd */
    float* save_rstd;
    const uint8_t* mask;   /* [R, C] keep flags or NULL */
    const float* resid;    /* highway input [R, C/2] or NULL */
    float* y;              /* [R, C] or [R, C/2] (highway) */
    float* ws;             /* mtts_bn_workspace_floats(C) floats */
    int R;
    int C;
    int training;
    float momentum;
    float eps;
    int act;
    float mask_scale;
    int hw_groups;         /* 0 plain; G > 0 highway with G groups, C = 2*G*Cg, gate chunk then value chunk per group */
    const float* dy;       /* bwd: [R, Cy] */
    float* dx;             /* bwd: [R, C] gradient w.r.t. the conv output */
    float* dgamma;
    float* dbeta;
    float* dresid;         /* bwd: [R, C/2] gradient w.r.t. the highway input (overwritten) or NULL */
} BnArgs;

long mtts_bn_workspace_floats(int C);
int mtts_bn_act_fwd(const BnArgs* args, void* stream);
int mtts_bn_act_bwd(const BnArgs* args, void* stream);

/* ---- skinny GEMM (batch rows x streamed weight rows) with fused epilogues ----------------------------------
 * Replaces torch.nn.LSTMCell (modules/layers.py:18-47, called at modules/tacotron2.py:185,188), the attention
 * query projection (modules/attention.py:68), per-step prenet (modules/tacotron2.py:37-46,181) and the
 * frame/stop projection (modules/tacotron2.py:192-193).  Y[B,N] = sum_s X_s[B,K_s] W_s[N,K_s]^T. */
typedef struct SkSeg {
    const float* x;
    const float* w;
    int K;
    int ldx;
    int ldw;
    int xpack;             /* 1: x is in the MFMA tile order written by the producer kernels (see mtts_pack_rows); 2: bf16 pair tiles
                              (SkinnyArgs.dg_pack_bf16 / mtts_pack_weight_bf16; plain products only, every operand of the launch, K % 32 == 0) */
    int wpack;             /* 1: w is in the MFMA tile order produced by mtts_pack_weight; 2: bf16 pair tiles (mtts_pack_weight_bf16) */
} SkSeg;

typedef struct SkinnyArgs {
    SkSeg seg[3];
    int nseg;
    int B;
    int N;
    int ksplit;            /* > 1: raw partial sums out[ks*out_ks + row*ldo + col] */
    float* out;
    int ldo;
    long out_ks;
    const float* bias;
    int act;
    const uint8_t* mask;
    int ldmask;
    float mask_scale;
    int lstm;              /* != 0: fused LSTM cell epilogue, N = 4H, gate order i,f,g,o */
    int H;
    const float* pre;      /* [B,4H] precomputed addend or NULL */
    int ldpre;
    const float* b_ih;
    const float* b_hh;
    const float* h_prev;
    const float* c_prev;
    float* h_out;
    float* c_out;
    float* gates_out;      /* [B,4H] activated gates saved for backward or NULL */
    const uint8_t* hmask;
    const uint8_t* cmask;
    float hscale;
    int zone;              /* 0 dropout on h (layers.py:44-47); 1 zoneout training; 2 zoneout eval (layers.py:26-34) */
    float zh;
    float zc;
    const int* lengths;    /* packed-sequence carry: rows with t >= lengths[row] keep their state (encoder.py:41-44) */
    int t;
    float* y_out;
    int ldy;
    /* lstm == 2: LSTM-cell BACKWARD epilogue.  The GEMM part (may be empty) yields one more addend of dh.
       dh[b,u] = gemm + dh_a + dh_b + sum_k part[k] ;  outputs dgates (pre-activation gate gradients), dc_out,
       and dh_carry_out (the part of dh that bypasses the cell: zoneout-kept or packed-sequence-carried rows). */
    const float* dh_a;     /* [B, ld_dh_a] addend or NULL */
    int ld_dh_a;
    const float* dh_b;     /* [B,H] addend (carry from the later step) or NULL */
    const float* part;     /* [n_part][B][part_ld] partial sums of the later step's input-gradient GEMM or NULL */
    int n_part;
    long part_ks;
    int part_ld;
    int part_col0;
    const float* gates;    /* [B,4H] activated gates saved by the forward */
    const float* dc_in;    /* [B,H] */
    float* dc_out;         /* [B,H] */
    float* dgates_out;     /* [B, ld_dgates] */
    int ld_dgates;
    float* dh_carry_out;   /* [B,H] or NULL */
    float* h_pack_out;     /* lstm == 1: additional copy of h_out in MFMA tile order (K = H) or NULL */
    float* dg_pack_out;    /* lstm == 2: additional copy of the gate gradients in MFMA tile order (K = 4H) or NULL */
    int dg_pack_bf16;      /* != 0: dg_pack_out receives bf16 pair tiles ([row tile][4H / 32][64 lanes][8 bf16], RNE) instead of fp32 tiles:
                              the x operand of the bf16 path's per-step input-gradient products (round 5) */
} SkinnyArgs;

int mtts_skinny_gemm(const SkinnyArgs* args, void* stream);
/* MFMA tile order ("packed"): a [rows, K] matrix (K % 16 == 0) is stored as [rows/16][K/16][64 lanes][4 floats]; lane
 * l = 16*q + i holds row 16*tile + i, columns 16*chunk + 4*q .. +3.  One wave instruction then fetches a whole 16x16 tile
 * as 1 KiB of contiguous memory directly in the v_mfma_f32_16x16x4_f32 operand layout.
 * mtts_pack_weight: rows are weight rows; lstm_H > 0 applies the fused-LSTM column order (tile cb = units 4cb..4cb+3 x 4 gates).
 * mtts_pack_rows: rows are batch rows (padded with zeros to a multiple of 16). */
int mtts_pack_weight(const float* src, int ld, int N, int K, int lstm_H, float* dst, void* stream);
int mtts_pack_rows(const float* src, int ld, int rows, int K, float* dst, void* stream);
/* bf16 pair tiles of a weight matrix (rows = weight rows, K % 32 == 0): [tiles of 16 rows][K / 32][64 lanes][16 B]; lane 16q + i holds
 * columns 32c + 4q .. + 3 and 32c + 16 + 4q .. + 3 of row 16 tile + i, RNE-rounded: the w operand (wpack == 2) of the per-step
 * input-gradient products in bf16 mode - two v_mfma_f32_16x16x16_bf16 per tile.  dst: N16 * K * 2 bytes (N16 = rows rounded up to 16). */
int mtts_pack_weight_bf16(const float* src, int ld, int N, int K, void* dst, void* stream);

/* ---- recurrent LSTM step without operand re-reads: K-split gate GEMM + (partial sum, cell, query partials) --------------
 * Replaces torch.nn.LSTMCell + dropout / zoneout (modules/layers.py:18-47, call site modules/tacotron2.py:185) and the
 * attention query projection of the SAME step (modules/attention.py:68) by two launches:
 *   G: P[ks][B][4H] = X[:, k-slice ks] W[:, k-slice ks]^T   (weights streamed once from the packed copy, X staged once per
 *      workgroup through LDS; 4H/128 column tiles x KS slices = 256 workgroups at H = 1024)
 *   C: gates = sum_ks P + pre + bias -> cell -> h, c, saved gates;  q_part[ut][B][A] = h[:, 16 ut..] W_q[:, 16 ut..]^T
 * Gate columns of P / pre / bias_u are UNIT-MAJOR: column 4u + g holds gate g (i,f,g,o) of unit u.
 * precision 0: fp32 operands, every product as six bf16 MFMA terms of exact 3-way splits (fp32-accurate);
 * precision 1: operands rounded to bf16 (weights stored as bf16 in the packed copy), fp32 accumulation and cell state;
 * precision 2 (batches above 64 rows only: the fused step kernel; the decoder uses it above 128 rows): fp32 operands, the weights stored as three pre-split bf16 planes
 *              (6 bytes per element), the same six-term products without a per-launch weight split.
 * Other forms behind the same entry (the launcher picks by shape): batches above 64 rows run the whole step as ONE launch (fused
 * kernels, no partial slabs); one or two rows at inference (gates_out, hmask, cmask NULL, zone != 1, precision 0, nb_max != 4) run a
 * GEMV-shaped launch on plain fp32 FMA (single-utterance synthesis, reference synthesize.py / Decoder.inference
 * modules/tacotron2.py:229-242).  `partials` may stay unused; results agree within fp32 rounding, not bit for bit across forms. */
typedef struct LstmPackArgs {
    const float* w[3];     /* up to 3 K-segments of the [4H, K_s] weight (row stride ldw[s]); K_s % 32 == 0 */
    int K[3];
    int ldw[3];
    int nseg;
    int H;
    int precision;
    void* dst;             /* mtts_lstm_packed_weight_bytes(H, sum K, precision) bytes */
    const float* b_ih;     /* optional: bias_u[4u + g] = b_ih[gH + u] + b_hh[gH + u] */
    const float* b_hh;
    float* bias_u;
} LstmPackArgs;

typedef struct LstmStepArgs {
    const float* x[3];     /* row-major [B, K_s] segments in the order of the packed weight */
    int K[3];
    int ldx[3];
    int nseg;
    const void* w_packed;
    int precision;
    int B;
    int H;
    float* partials;       /* mtts_lstm_step_partial_floats(B, H, sum K) floats */
    const float* pre;      /* unit-major [B, 4H] addend (hoisted input projection) or NULL */
    int ldpre;
    const float* bias_u;   /* unit-major [4H] or NULL */
    const float* h_prev;   /* [B,H] (zoneout only) */
    const float* c_prev;
    float* h_out;
    float* c_out;
    float* gates_out;      /* [B,4H] activated gates, GATE-major (column gH + u) like mtts_skinny_gemm's, or NULL */
    const uint8_t* hmask;
    const uint8_t* cmask;
    float hscale;
    int zone;              /* 0 dropout on h; 1 zoneout training; 2 zoneout eval */
    float zh;
    float zc;
    const float* w_query;  /* [A, H] row-major or NULL */
    int A;
    float* qpart;          /* [H/16][B][A] query-projection partials (summed by mtts_attn_step_fwd with kq = H/16) or NULL */
    int nb_max;            /* 32-wide k-blocks per K-slice: 0 = up to 10 (fewest partial slabs); 4 = short slices, two workgroups per CU
                              (training: shares CUs with concurrently running GEMM workgroups).  Batches above 64 rows: 4 = another step
                              kernel runs beside this one (keeps the two-workgroups-per-CU kernel), 0 = the launch has the chip to itself */
} LstmStepArgs;

int mtts_lstm_step_ksplit(int k_total);                    /* upper bound of the slab count over every nb_max >= 4 */
long mtts_lstm_step_partial_floats(int B, int H, int k_total);
long mtts_lstm_packed_weight_bytes(int H, int k_total, int precision);
int mtts_lstm_pack_weights(const LstmPackArgs* args, void* stream);
/* dst[(4u + g) K + k] = src[(gH + u) ld + k]: rows of a [4H, K] LSTM matrix into unit-major order (for the hoisted projection) */
int mtts_lstm_rows_unit_major(const float* src, int ld, int H, int K, float* dst, void* stream);
int mtts_lstm_step_fwd(const LstmStepArgs* args, void* stream);

/* ---- location-sensitive attention step -------------------------------------------------------------------
 * Replaces LocationSensitiveAttention.forward for one decoder step: modules/attention.py:39-45,67-86.
 * The location Conv1d(1->C,k) followed by Linear(C->A) is applied as ONE k-tap filter bank U = W_loc * W_conv
 * ([A,k]); PL = M + bias + loc(cum) is produced for the NEXT step by the current one (it only depends on the
 * cumulative alignment), so the per-step critical path is energies -> masked softmax -> context. */
typedef struct AttnStepArgs {
    const float* qpart;    /* [kq][B][A] partial query projections (summed here) */
    int kq;
    long q_ks;
    const float* PL;       /* [B,L,A] = M + bias + loc(cum_in).  TWO FORMS (pick with mtts_attn_step_form and do not mix them between the
                              steps of one decode): form 0 (B < 128, or a shape the large-batch kernel does not take) READS PL and WRITES
                              PL_next; form 1 (attn_step_big_kernel: B >= 128, nch 1 or 2) IGNORES PL, recomputes it
                              in registers from Mt / bias / U / cum_in and does NOT write PL_next - Mt, U and bias are then required. */
    float* PL_next;        /* [B,L,A] for cum_out (nullable; form 0 only) */
    const float* Mt;       /* [B,L,A] memory transform (form 1: required) */
    const float* U;        /* [A,ksz] */
    const float* bias;     /* [A] */
    const float* v;        /* [A] energy weights */
    const float* memory;   /* [B,L,Dm] */
    const int* lengths;    /* [B] */
    const float* cum_in;   /* [B,L] */
    float* cum_out;        /* [B,L] */
    float* w_out;          /* [B,L] alignment of this step */
    float* ctx_out;        /* [B,Dm] */
    float* q_out;          /* [B,A] summed query projection saved for backward (nullable) */
    float* ctx_pack_out;   /* context in MFMA tile order (K = Dm, Dm % 16 == 0) or NULL */
    int B;
    int L;
    int A;
    int Dm;
    int ksz;
    int nch;               /* workgroups per sample */
} AttnStepArgs;

int mtts_attn_step_fwd(const AttnStepArgs* args, void* stream);
/* 1 when mtts_attn_step_fwd runs the large-batch kernel for this shape (PL unused, PL_next not written), else 0 */
int mtts_attn_step_form(int B, int L, int A, int Dm, int ksz, int kq, int nch);

/* ---- whole decoder loop, forward ----------------------------------------------------------------------------
 * Replaces Decoder._decode (modules/tacotron2.py:148-209) incl. _target_init (:126-133), the attention reset
 * (modules/attention.py:23-28) and both regularised LSTM cells.  Time-major saved state so that step slices are
 * contiguous.  Steps [t0, t1) are executed; state arrays are indexed by absolute step, so a call can resume. */
typedef struct DecoderArgs {
    int B;
    int L;
    int T;                 /* allocated steps */
    int t0;
    int t1;
    int M;                 /* num_mels */
    int P;                 /* prenet width */
    int H;
    int A;
    int Dm;
    int ksz;
    int C;                 /* location channels */
    int n_prenet;          /* prenet layers (<= 4) */
    int training;
    int zone;              /* decoder_regularization: 0 dropout, 1 zoneout */
    float p_prenet;        /* dropout prob of the prenet (always on) */
    float p_hidden;        /* dropout_hidden, or zoneout_hidden */
    float p_cell;          /* zoneout_cell */
    /* inputs */
    const float* memory;   /* [B,L,Dm] encoder output (+ embeddings) */
    const int* lengths;    /* [B] */
    const float* frames_in;/* [T,B,M] teacher frames (frame fed at step t), NULL when free running */
    const uint8_t* teacher;/* HOST array [T]: 1 = feed frames_in[t], 0 = feed own prediction (modules/tacotron2.py:171,181) */
    /* weights */
    const float* prenet_w[4];
    const float* prenet_b[4];
    const float* att_w_ih; /* [4H, P+Dm] */
    const float* att_w_hh; /* [4H, H] */
    const float* att_b_ih;
    const float* att_b_hh;
    const float* gen_w_ih; /* [4H, H+Dm] */
    const float* gen_w_hh;
    const float* gen_b_ih;
    const float* gen_b_hh;
    const float* w_query;  /* [A,H] */
    const float* w_memory; /* [A,Dm] */
    const float* w_loc;    /* [A,C] */
    const float* w_conv;   /* [C,ksz] */
    const float* att_bias; /* [A] */
    const float* w_energy; /* [A] */
    const float* w_out;    /* [M+1, H+Dm] frame rows then the stop row */
    const float* b_out;    /* [M+1] */
    /* dropout / zoneout keep flags (NULL = none) */
    const uint8_t* prenet_mask[4];  /* teacher path: [T,B,P] each; free-running steps index the same arrays by t */
    const uint8_t* att_hmask;       /* [T,B,H] */
    const uint8_t* att_cmask;
    const uint8_t* gen_hmask;
    const uint8_t* gen_cmask;
    /* saved state / outputs (time-major) */
    float* prenet_act[4];  /* [T,B,P] activations of every prenet layer (last = LSTM input) */
    float* U;              /* [A,ksz] */
    float* Mt;             /* [B,L,A] */
    float* PL;             /* [2,B,L,A] ping-pong */
    float* qpart;          /* [kq,B,A] */
    int kq;
    float* h_att;          /* [T+1,B,H], slot 0 = initial state */
    float* c_att;
    float* h_gen;
    float* c_gen;
    float* ctx;            /* [T+1,B,Dm] */
    float* cum;            /* [T+1,B,L] */
    float* align;          /* [T,B,L] */
    float* gates_att;      /* [T,B,4H] or NULL */
    float* gates_gen;
    float* out;            /* [T+1,B,Mo] (Mo = M+1 rounded up to 4): slot t+1 = frame + stop logit of step t, slot 0 = zero frame */
    float* pre_att;        /* [T,B,4H] hoisted input projection workspace (fast path) or NULL */
    float* pre_gen;        /* [T,B,4H] */
    float* q_all;          /* [T,B,A] query projections saved for backward or NULL */
    /* optional MFMA-tile-order copies used by the step kernels (all NULL -> row-major operands); Bp = B rounded up to 16 */
    float* h_att_p;        /* [T+1][Bp*H] */
    float* h_gen_p;        /* [T+1][Bp*H] */
    float* ctx_p;          /* [T+1][Bp*Dm] */
    float* att_w_ctx_p;    /* packed W_ih[:, P:]  (4H x Dm, LSTM column order) */
    float* att_w_hh_p;     /* packed W_hh         (4H x H) */
    float* gen_w_hh_p;
    float* w_query_p;      /* packed W_query      (A x H) */
    int fast;              /* 1: all steps teacher forced -> hoisted projections + deferred generator chain */
    /* optional K-split step path of the attention LSTM (fast schedule only; see LstmStepArgs).  All NULL -> skinny kernels.
       Needs Dm % 32 == 0, H % 32 == 0, A % 16 == 0 and qpart sized [max(kq, H/16)][B][A]. */
    void* att_w2p;         /* mtts_lstm_packed_weight_bytes(H, Dm + H, precision) bytes: packed [W_ih[:, P:] | W_hh] */
    float* att_bias_u;     /* [4H] unit-major b_ih + b_hh */
    float* att_w_pre_u;    /* [4H, P] rows of W_ih[:, :P] in unit-major order (pre_att is then unit-major) */
    float* gate_part;      /* mtts_lstm_step_partial_floats(B, H, Dm + H) floats */
    /* the same for the generator LSTM's recurrent step (its [h_att, ctx] input projection is hoisted per chunk) */
    void* gen_w2p;         /* mtts_lstm_packed_weight_bytes(H, H, precision) bytes: packed W_hh */
    float* gen_bias_u;     /* [4H] */
    float* gen_w_ih_u;     /* [4H, H + Dm] rows of W_ih in unit-major order (pre_gen is then unit-major) */
    float* gate_part_gen;  /* mtts_lstm_step_partial_floats(B, H, H) floats */
    float* prenet_wp[2];   /* free-running steps: MFMA-tile-order copies of the two prenet weights ([P, M] and [P, P]; M, P % 16 == 0) */
    int precision;         /* 0: fp32 (3-way bf16 split products); 1: bf16 operands in the step GEMMs */
    /* exchange / synchronisation workspace of the persistent recurrence kernels (csrc/persist.hip): zero-initialised by the caller
       once, mtts_decoder_persist_ws_bytes(B, L, H, Dm, A) bytes; NULL -> per-step launches everywhere */
    void* persist_ws;
    long persist_ws_bytes;
    int* persist_err;      /* optional long-lived device error word of the persistent kernels (0 = ok, 2 = a grid barrier gave up);
                              NULL -> the word inside persist_ws (mtts_decoder_persist_status) */
} DecoderArgs;

int mtts_decoder_fwd(const DecoderArgs* args, void* stream);
/* hipGraph form of a FREE-RUNNING range [t0, t1) (general schedule, no teacher frames): the launches are captured once per distinct
 * argument block - pointers, sizes and step range all take part - and replayed with one hipGraphLaunch afterwards (first call eager,
 * second call captures).  The caller keeps every buffer alive at the same address between calls; `stream` must be a non-default
 * stream.  *replayed (nullable) = 1 when a graph ran.  Replaces the per-step Python / kernel-launch loop of
 * Decoder.inference (modules/tacotron2.py:216-219,178-207) for BASELINE configs[4]. */
int mtts_decoder_fwd_graphed(const DecoderArgs* args, void* stream, int* replayed);
/* Destroys the graphs captured for `stream` (call when the fixed-address buffers they were captured over are retired). */
int mtts_decoder_graphs_clear(void* stream);
/* Stop rule of batched free-running synthesis evaluated ON THE DEVICE (reference Decoder._decode, modules/tacotron2.py:201-207, per
 * sample): walks the stop logits of frames [t0, t1) in `out` ([T+1][B][Mo], column M of slot t+1 = frame t) and updates
 * state[0..B) = armed counters (-1 = not armed), state[B..2B) = frame count at which the utterance ended (-1 = running);
 * *running (device int, one word per in-flight chunk) = utterances still running after this chunk.  The caller fills `state` with
 * -1 before the first chunk and reads ONE int per chunk instead of copying every stop logit to the host.
 * stop_threshold = probability (>= 1: never stop). */
int mtts_stop_rule_update(const float* out, int t0, int t1, int B, int Mo, int M, float stop_threshold, int stop_frames, int* state,
                          int* running, void* stream);
/* Persistent (weights-stationary, one launch for all steps) recurrences of the teacher-forced schedule (csrc/persist.hip; replace the
 * per-step LSTMCell / attention launches of Decoder._decode, modules/tacotron2.py:180-193).  mtts_decoder_fwd uses them when
 * DecoderArgs.persist_ws is set, the shape fits (H = 1024, B <= 64, fp32) and MTTS_PERSIST != 0.
 * mtts_decoder_persist_ws_bytes: size of that workspace (the caller zero-fills it once after allocating it).
 * mtts_decoder_persist_status: synchronises `stream` and returns the device error word of the workspace
 * (0 = ok, 2 = a grid barrier gave up; results of that decode are invalid), -1 on a runtime error. */
long mtts_decoder_persist_ws_bytes(int B, int L, int H, int Dm, int A);
int mtts_decoder_persist_status(const void* persist_ws, void* stream);

/* ---- bidirectional LSTM over padded batch with packed-sequence semantics ----------------------------------
 * Replaces nn.LSTM(bidirectional) + pack/pad of modules/encoder.py:41-44. */
typedef struct BiLstmArgs {
    int B;
    int L;
    int Cin;
    int H;                 /* per direction */
    const float* x;        /* [L,B,Cin] time-major */
    const int* lengths;
    const float* w_ih[2];  /* [4H,Cin] fwd, reverse */
    const float* w_hh[2];  /* [4H,H] */
    const float* b_ih[2];
    const float* b_hh[2];
    float* xproj[2];       /* [L,B,4H] workspace */
    float* h[2];           /* [L+1,B,H]; fwd: slot t -> t+1; reverse: slot t+1 -> t (slot L zero) */
    float* c[2];
    float* gates[2];       /* [L,B,4H] or NULL */
    float* y;              /* [B,L,2H] */
} BiLstmArgs;

int mtts_bilstm_fwd(const BiLstmArgs* args, void* stream);

/* ---- attention step, backward ---------------------------------------------------------------------------------
 * Gradient of one LocationSensitiveAttention step (modules/attention.py:39-45,67-86).  PL is recomputed from the
 * saved cumulative alignment; per-workgroup slabs (dU, dv, dbias) avoid global atomics on shared parameters. */
typedef struct AttnBwdArgs {
    const float* q;        /* [B,A] saved query projection of this step */
    const float* Mt;       /* [B,L,A] */
    const float* U;        /* [A,ksz] */
    const float* bias;
    const float* v;
    const float* memory;   /* [B,L,Dm] */
    const float* ctx;      /* [B,Dm] context produced by this step */
    const int* lengths;
    const float* w;        /* [B,L] alignment of this step */
    const float* cum_in;   /* [B,L] cumulative alignment fed to this step */
    const float* dalign;   /* [B,L] external gradient w.r.t. the alignment or NULL */
    const float* dcum_out; /* [B,L] gradient w.r.t. the cumulative alignment AFTER this step */
    float* dcum_in;        /* [B,L] gradient w.r.t. cum_in (zero-initialised by the caller; accumulated atomically) */
    const float* dctx;     /* [B,Dm] direct part of the context gradient */
    float* dctx_total;     /* [B,Dm] out: direct + sum of partials (kept for the memory-gradient GEMM) */
    const float* part;     /* [n_part][B][part_ld] partials carrying the remaining context gradient (columns [0,Dm)) or NULL */
    int n_part;
    long part_ks;
    int part_ld;
    float* dq;             /* [B,A] zero-initialised, accumulated atomically */
    float* dMt;            /* [B,L,A] accumulated (+=) */
    float* dU_slab;        /* [B*nch][A*ksz] accumulated (+=) */
    float* dv_slab;        /* [B*nch][A] */
    float* dbias_slab;     /* [B*nch][A] */
    int B;
    int L;
    int A;
    int Dm;
    int ksz;
    int nch;
} AttnBwdArgs;

int mtts_attn_step_bwd(const AttnBwdArgs* args, void* stream);

/* ---- whole decoder loop, backward (BPTT) -------------------------------------------------------------------------
 * Consumes the buffers saved by mtts_decoder_fwd (same DecoderArgs) plus the gradients of its outputs; all weight
 * gradients are formed by large MFMA GEMMs over the saved per-step gate gradients after the sequential sweeps. */
typedef struct DecoderGradArgs {
    const float* dout;     /* [T+1,B,Mo] slot t+1 = gradient of (frame, stop) of step t */
    const float* dalign;   /* [T,B,L] or NULL */
    /* workspaces */
    float* att_w_rec_T;    /* [Dm+H,4H] */
    float* att_w_ih_T;     /* general schedule: [P+Dm+H,4H] = [W_ih | W_hh]^T of the attention LSTM */
    float* gen_w_hh_T;     /* [H,4H] */
    float* gen_w_ih_T;     /* general schedule: [H+Dm+H,4H] = [W_ih | W_hh]^T of the generator LSTM */
    float* w_out_T;        /* general schedule: [H+Dm,Mo] transposed frame/stop projection (zero padded columns) */
    float* prenet_w_T[4];  /* general schedule: transposed prenet weights [in_i, P] */
    float* step_ws;        /* general schedule: B*(P+Dm+H + 2H+Dm + H+Dm + M) floats of per-step scratch */
    float* frames_fed;     /* general schedule: [T,B,M] frame actually fed to the prenet at each step */
    float* w_query_T;      /* [H,A] */
    float* dG_att;         /* [T,B,4H] */
    float* dG_gen;         /* [T,B,4H] */
    float* dG_att_p;       /* [T][Bp*4H] MFMA tile order copies (optional) */
    float* dG_gen_p;
    float* att_w_rec_Tp;   /* packed [Dm+H, 4H] */
    float* gen_w_hh_Tp;    /* packed [H, 4H] */
    float* dHG;            /* [T,B,H] */
    float* dHA;            /* [T,B,H] */
    float* dctx_all;       /* [T+1,B,Dm] direct (batched) parts, slot t+1 <-> context of step t */
    float* dctx_tot;       /* [T+1,B,Dm] totals */
    float* dcum_all;       /* [T+1,B,L] zero-initialised */
    float* dq_all;         /* [T,B,A] zero-initialised */
    float* part_gen;       /* [ksb,B,H] */
    float* part_att;       /* [ksb_ctx,B,Dm] followed by [ksb,B,H] */
    int ksb;
    int ksb_ctx;           /* K-splits of the context-column input-gradient GEMM (0: same as ksb) */
    float* dc_att;         /* [2,B,H] zero-initialised */
    float* dc_gen;         /* [2,B,H] zero-initialised */
    float* dh_carry_att;   /* [2,B,H] zero-initialised: part of dh that bypasses the cell (zoneout) */
    float* dh_carry_gen;   /* [2,B,H] zero-initialised */
    float* dMt;            /* [B,L,A] zero-initialised */
    float* dU_slab;        /* [B*nch,A*ksz] zero-initialised */
    float* dv_slab;        /* [B*nch,A] zero-initialised */
    float* dbias_slab;     /* [B*nch,A] zero-initialised */
    int nch;
    float* dU;             /* [A,ksz] */
    float* dpren;          /* [n_prenet][T,B,P] gradient scratch for the prenet layers */
    float* colsum_ws;      /* mtts_colsum_workspace_floats(max C) floats */
    float* part_ring;      /* persistent backward (round 6, csrc/pbwd.hip): [part_ring_slots][ksb_ctx*B*Dm + ksb*B*H] partial slabs indexed by
                              the step (slot t % part_ring_slots), or NULL: the per-step launch schedule runs */
    int part_ring_slots;   /* >= mtts_decoder_bwd_ring_slots() */
    /* outputs */
    float* dmemory;        /* [B,L,Dm] */
    float* d_prenet_w[4];
    float* d_prenet_b[4];
    float* d_att_w_ih;
    float* d_att_w_hh;
    float* d_att_b_ih;
    float* d_att_b_hh;
    float* d_gen_w_ih;
    float* d_gen_w_hh;
    float* d_gen_b_ih;
    float* d_gen_b_hh;
    float* d_w_query;
    float* d_w_memory;
    float* d_w_loc;
    float* d_w_conv;
    float* d_att_bias;
    float* d_w_energy;
    float* d_w_out;
    float* d_b_out;
} DecoderGradArgs;

int mtts_decoder_bwd(const DecoderArgs* fwd, const DecoderGradArgs* grad, void* stream);
/* slots DecoderGradArgs.part_ring needs for the persistent backward at the current chunk length (chunk + 2) */
int mtts_decoder_bwd_ring_slots(void);

/* Backward of mtts_bilstm_fwd. */
typedef struct BiLstmGradArgs {
    const float* dy;       /* [B,L,2H] */
    float* w_hh_T[2];      /* [H,4H] workspace */
    float* dxproj[2];      /* [L,B,4H] gate gradients */
    float* part;           /* [2 directions][ksb,B,H] (the two directions run concurrently on two streams) */
    int ksb;
    float* dc;             /* [2 directions][2,B,H] */
    float* dh_carry;       /* [2 directions][2,B,H] */
    float* colsum_ws;
    float* dx;             /* [L,B,Cin] time-major */
    float* d_w_ih[2];
    float* d_w_hh[2];
    float* d_b_ih[2];
    float* d_b_hh[2];
} BiLstmGradArgs;

int mtts_bilstm_bwd(const BiLstmArgs* fwd, const BiLstmGradArgs* grad, void* stream);

/* ---- training loss and optimizer step -----------------------------------------------------
 * mtts_tacotron_loss: TacotronLoss.forward (modules/tacotron2.py:443-485) without the classifier term: values in out[0..4]
 * (mel_pre, mel_pos, stop_token, guided_att, total) and the gradients w.r.t. pre/post/stop/alignment scaled by gscale. */
typedef struct TacoLossArgs {
    const float* pre;          /* [B,M,T] */
    const float* post;         /* [B,M,T] */
    const float* target;       /* [B,M,T] */
    const float* stop;         /* [B,T] logits */
    const float* stop_target;  /* [B,T] */
    const float* align;        /* [B,T,L] or NULL */
    const int* text_len;       /* [B] */
    const int* target_len;     /* [B] */
    float* d_pre;
    float* d_post;
    float* d_stop;
    float* d_align;
    float* partials;           /* [nblk*4] */
    float* out;                /* [5] */
    int B;
    int M;
    int T;
    int L;
    int nblk;
    int ga_on;                 /* guided attention active (guided_att_steps > 0) */
    float g;                   /* guided attention tolerance */
    float pos_weight;          /* 100 in the reference */
    float gscale;              /* upstream gradient of the total loss (1 for loss.backward()) */
    const float* post_target;  /* [B,M,T] target of the post-net output when it differs from `target` (NULL: the same tensor) */
} TacoLossArgs;

int mtts_tacotron_loss(const TacoLossArgs* args, void* stream);

/* mtts_masked_cross_entropy: ReversalClassifier.loss (modules/classifier.py:62-69) times `scale`: cross entropy of pred [B,L,S] against
 * speakers[b] at every character l < lengths[b] (the rest is ignore_index), mean over the valid characters.  row_loss [B*L] holds the
 * per-row contributions (sum them with mtts_colsum), dpred [B,L,S] the gradient of the scaled mean. */
int mtts_masked_cross_entropy(const float* pred, const int64_t* speakers, const int* lengths, float* row_loss, float* dpred,
                              int B, int L, int S, float scale, void* stream);

/* mtts_clip_adam_step: torch.nn.utils.clip_grad_norm_(params, max_norm) followed by torch.optim.Adam.step() with L2-coupled
 * weight decay (train.py:84-85,260-270).  ptrs: device array of 4 pointers per tensor {param, grad, exp_avg, exp_avg_sq};
 * chunks partition every tensor into blocks of work. */
typedef struct AdamArgs {
    const int64_t* ptrs;
    const int* chunk_tensor;
    const int64_t* chunk_off;
    const int* chunk_len;
    float* norm_partials;      /* [nchunks] */
    float* norm_out;           /* [2]: total gradient norm, clip coefficient */
    int nchunks;
    float max_norm;            /* <= 0: no clipping */
    float weight_decay;
    float beta1;
    float beta2;
    float eps;
    float step_size;           /* lr / (1 - beta1^t) */
    float inv_sqrt_bc2;        /* 1 / sqrt(1 - beta2^t) */
    const int* guard;          /* NULL, or the device error words of this GPU (int[2]: [0] invalid input seen by a kernel, [1] a
                                  persistent decoder kernel gave up on a hand-off, DecoderArgs.persist_err).  With a guard the update
                                  is SKIPPED on the device (no host synchronisation) when guard[1] != 0 or the gradient norm is not
                                  finite: the step whose decode was invalid never reaches the weights; norm_out[1] is then -1 */
    int phase;                 /* 0: norm + update over this table; 1: only the global gradient norm / clip coefficient into norm_out
                                  (table = ALL parameters); 2: only the update, with the clip coefficient already in norm_out[1]
                                  (table = one parameter group / bias-correction step: train.py:261-270 trains the encoder with
                                  its own learning rate while clip_grad_norm_ still spans every parameter) */
} AdamArgs;

int mtts_clip_adam_step(const AdamArgs* args, void* stream);

/* ---- parameter generator of the 'generated' encoder (K2) ---------------------------------------------------------------------
 * Replaces the second Linear of Conv1dGenerated + the .view + the layout change in front of F.conv1d (modules/generated.py:34-42):
 * w_packed[(g Og + o), t, c] = b_kernel[j] + sum_b hidden[g, b] w_kernel[j, b],  j = (o Cg + c) k + t   (the implicit-GEMM layout
 * [O, k, I/G] that mtts_gemm_ex consumes), and backwards from the conv's packed weight gradient:
 * d_w_kernel[j, b] = sum_g d_w_packed[g, j] hidden[g, b];  d_b_kernel[j] = sum_g d_w_packed[g, j];
 * d_hidden[g, b] = sum_j d_w_packed[g, j] w_kernel[j, b] as per-workgroup partial rows (sum them with mtts_colsum). */
typedef struct GenParamsArgs {
    const float* hidden;        /* [G, bott] output of the bottleneck Linear */
    const float* w_kernel;      /* [Og*Cg*k, bott] */
    const float* b_kernel;      /* [Og*Cg*k] or NULL */
    float* w_packed;            /* fwd out: [G*Og, k, Cg] */
    const float* d_w_packed;    /* bwd in */
    float* d_w_kernel;          /* bwd out */
    float* d_b_kernel;          /* bwd out or NULL */
    float* d_hidden_slab;       /* bwd out: [mtts_gen_params_slabs(Og, Cg)][G*bott] */
    int G;
    int bott;
    int Og;
    int Cg;
    int k;
} GenParamsArgs;

long mtts_gen_params_slabs(int Og, int Cg);
int mtts_gen_params_fwd(const GenParamsArgs* args, void* stream);
int mtts_gen_params_bwd(const GenParamsArgs* args, void* stream);

/* ---- small data-movement kernels --------------------------------------------------------------------------- */
/* Embedding lookup (modules/tacotron2.py:363, :122 speaker/language tables): out[r, col0:col0+D] = table[ids[r]].
 * `vocab` = rows of the table.  An id outside [0, vocab) reads as a zero row, never touches memory and sets *err (device int,
 * nullable) to 1; the caller checks it when it next synchronises (torch.nn.Embedding device-asserts in that case). */
int mtts_embedding_fwd(const float* table, const int64_t* ids, float* out, int rows, int D, int ldo, int col0, long vocab,
                       int* err, void* stream);
/* dtable[ids[r]] += dout[r, col0:col0+D]; rows with ids == padding_idx (padding_idx < 0: none) or outside [0, vocab) are skipped */
int mtts_embedding_bwd(const float* dout, const int64_t* ids, float* dtable, int rows, int D, int ldo, int col0,
                       int padding_idx, long vocab, void* stream);
/* out[r*ldo + c] = in[r*ldi + c] for c < cols (strided 2-D copy, used for concatenations) */
int mtts_copy2d(const float* in, float* out, int rows, int cols, int ldi, int ldo, void* stream);
/* [O, I, k] <-> [O, k, I] weight repack for the implicit-GEMM convolution (to_packed != 0: torch layout -> packed) */
int mtts_conv_weight_pack(const float* in, float* out, int O, int I, int k, int to_packed, void* stream);

/* out[c, r] = in[r, c]  (in [rows, cols] row-major) */
int mtts_transpose(const float* in, float* out, int rows, int cols, void* stream);
/* out[c] = sum_r x[r*ld + c], deterministic two-stage reduction; ws: mtts_colsum_workspace_floats(cols) floats */
long mtts_colsum_workspace_floats(int cols);
int mtts_colsum(const float* x, float* out, int rows, int cols, int ld, float* ws, void* stream);
/* dz = (y > 0) ? dy * scale : 0 -- backward of ReLU followed by dropout, y being the dropped activation */
int mtts_relu_mask_bwd(const float* dy, const float* y, float* dz, long n, float scale, void* stream);

/* Gradient reversal backward (modules/classifier.py:16-18): out = clamp(g, -c, c) * (-l) */
int mtts_grad_reverse_clamp(const float* g, float* out, long n, float l, float c, void* stream);

/* Sampling of the dominant step kernel (attention-LSTM skinny GEMM) with HIP events on the launch stream:
 * begin() creates `max_samples` event pairs and samples every `stride`-th decoder step; end() synchronises
 * them and returns the summed duration in ms and the sample count.  Not for use inside timed regions' setup. */
/* uint8 keep flags (1 = keep) with P(keep) = 1 - p from Philox4x32-10 keyed by `seed`; element i uses counter offset + i/8.
 * Draws of consecutive calls stay independent when the caller advances `offset` by (n + 7) / 8.  Replaces
 * `torch.rand(shape) >= p` at every dropout / zoneout site (modules/layers.py:27,37-40,82-86; modules/tacotron2.py:44). */
int mtts_dropout_keep_mask(uint8_t* out, long n, float p, uint64_t seed, uint64_t offset, void* stream);

int mtts_prof_begin(int max_samples, int stride);
/* Launches the no-op `mtts_marker_kernel` on `stream`: region boundaries for rocprofv3 traces / PMC passes (bench.py). */
int mtts_prof_marker(int tag, void* stream);
int mtts_prof_end(float* total_ms, int* count);
/* Test hook: launches `workgroups` x 64 threads that each hold `lds_bytes` of LDS and sleep for `ms` milliseconds on `stream` - a
 * foreign resident kernel beside which the persistent decoder kernels must start late instead of timing out (tests/test_gpu_persist.py). */
int mtts_debug_occupy(int workgroups, int lds_bytes, float ms, void* stream);
/* summed duration (ms) of the EMPTY event brackets recorded in front of every sample: the cost of an event pair with nothing
 * between, subtracted by bench.py from the kernel brackets */
float mtts_prof_empty_ms(void);

/* launches of the pre-split GEMM core (pack passes + gemm_planes_kernel, csrc/gemm_planes.h) in this process: mtts_gemm_ex takes it
 * for plain GEMMs above a size threshold (MTTS_GEMM_PLANES=0: never; MTTS_PLANES_MIN_GFLOP / _BF16: the threshold).  Tests use the
 * count to see which core ran. */
long mtts_gemm_planes_count(void);

const char* mtts_last_error(void);
/* ABI version.  102 (round 6): DecoderGradArgs gained part_ring / part_ring_slots; new exports mtts_decoder_bwd_ring_slots,
 * mtts_set_planes_workspace, mtts_planes_trim, mtts_debug_occupy.  101 (round 5) is NOT layout-compatible with 100: AdamArgs gained `guard`, LstmPackArgs lost `plain_rows`, AttnBwdArgs
 * lost `hsum_out` / `hsum_cols`, DecoderGradArgs lost three fields, the mtts_ksplit_* exports are gone, LstmStepArgs.precision takes 2
 * (pre-split planes), DecoderArgs gained the long-input fields of the persistent decoder.  Bindings that do not generate their structs
 * from this header must check mtts_version() and mtts_sizeof_struct(). */
int mtts_version(void);
/* Bitmask of compile-time switches that make a build compute WRONG results on purpose (timing experiments).  The product sources
 * have none left (round 4); always 0 - bindings keep refusing anything else. */
int mtts_build_flags(void);
/* sizeof() of the structs above in declaration order (0 = GemmArgs ... 6 = BiLstmArgs, 7 = AttnBwdArgs, 8 = DecoderGradArgs, 9 = BiLstmGradArgs, 10 = TacoLossArgs, 11 = AdamArgs, 12 = LstmPackArgs, 13 = LstmStepArgs, 14 = GenParamsArgs); -1 when out of range */
int mtts_sizeof_struct(int which);


/* ---- buffer-size queries -----------------------------------------------------------------------------------------------------
 * ELEMENTS the caller must allocate for a caller-provided buffer of the argument blocks, by FIELD NAME (the struct member's
 * name; per-layer arrays: "prenet_act", "prenet_mask", "prenet_wp0", "prenet_wp1", "prenet_w_T0" (first layer) / "prenet_w_T").
 * Fill the shape fields first (DecoderArgs: B, L, T, M, P, H, A, Dm, ksz, C, n_prenet, kq, fast, precision; DecoderGradArgs: ksb,
 * ksb_ctx, nch; BiLstmArgs: B, L, Cin, H).  Element = float, except att_w2p / gen_w2p / persist_ws (bytes) and the
 * keep-flag masks (uint8).  -1 = unknown field.  Replaces the size formulas a binding would otherwise copy from the comments. */
long mtts_decoder_buffer_elems(const DecoderArgs* args, const char* field);
long mtts_decoder_grad_buffer_elems(const DecoderArgs* fwd, const DecoderGradArgs* grad, const char* field);
long mtts_bilstm_buffer_elems(const BiLstmArgs* args, int ksb, const char* field);

#ifdef __cplusplus
}
#endif
#endif
