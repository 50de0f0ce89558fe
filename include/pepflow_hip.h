/*
 * pepflow_hip.h -- C ABI of libpepflow_hip.so: hand-written gfx950 (MI355X) kernels for the
 * PepFlow multi-modal flow-matching denoise path.
 *
 * The reference (Ced3-han/PepFlowww) has NO native/FFI interface for this path: it sits behind
 * a plain torch nn.Module API.  This header therefore DEFINES the boundary a
 * replacement must export; each entry point names the reference code it replaces
 * (paths relative to /root/reference).  The reference-side binding (ctypes) is shown in
 * INTEGRATION.md and implemented in pepflowww_amd/_capi.py.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch allocator); kernels never
 *     allocate, free or keep state; weights are read-only;
 *   - all float tensors are fp32, contiguous row-major unless a leading dimension (ld*) is given;
 *     sequences are int64; masks are fp32 0/1 ([B*L]);
 *   - launches go to the hipStream_t passed as `stream` (opaque void*), no device sync;
 *   - return value: hipError_t as int (0 = hipSuccess), PF_E_* (negative) for argument errors;
 *     nothing throws.
 */
#ifndef PEPFLOW_HIP_H
#define PEPFLOW_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PF_ABI_VERSION 59
#define PF_ATT_VROWS 164             /* rows of a head's transposed value block: 128 channels + 12 points x 3 */
/* att_vt (f16 mode, ABI 53): a head's transposed values [PF_ATT_VROWS rows][keys] in the FRAGMENT ORDER of the score kernel's second
 * product -- block (tile n, 32-key step) = 512 f16 = the eight operand slots of each of its 64 lanes: row c sits in tile n = c & 7 as
 * lane row r = c >> 3 (channels), or tile 8 + (c - 128) / 16 as r = (c - 128) % 16 (point coordinates); key j in step j / 32 at
 * K group (j / 8) % 4, slot j % 8.  PF_ATT_VT_HEAD(L) f16 per (sample, head); PF_ATT_VT_OFF(c, j, L) the offset of an element. */
#define PF_ATT_VT_NST(L) (((L) + 31) >> 5)
#define PF_ATT_VT_HEAD(L) ((size_t)11 * PF_ATT_VT_NST(L) * 512)
#define PF_ATT_VT_OFF(c, j, L) ((((size_t)(((c) < 128 ? ((c) & 7) : 8 + (((c) - 128) >> 4)) * PF_ATT_VT_NST(L) + ((j) >> 5)) * 64 + (((j) >> 3) & 3) * 16 + \
                                  ((c) < 128 ? ((c) >> 3) : (((c) - 128) & 15))) * 8) + ((j) & 7))
#define PF_E_BADARG (-1)
#define PF_E_TOOLARGE (-2)

typedef void* pf_stream_t;

/* model constants (configs/learn_angle.yaml:3-14) baked into the kernels */
#define PF_C_S 128
#define PF_C_Z 64
#define PF_HEADS 8
#define PF_C_HID 128
#define PF_QK_PTS 8
#define PF_V_PTS 12
#define PF_IPA_PROJ 3744  /* q 1024 | kv 2048 | q_pts 192 | kv_pts 480 */
#define PF_IPA_FEATS 1536 /* o 1024 | o_pt x,y,z 3*96 | |o_pt| 96 | o_pair 128 */
#define PF_ET_PRE 512     /* a 192 | c 192 | d 64 | e 64 */

int pf_abi_version(void);

/* MFMA fragment-layout self test: C[16,16] = A[16,K] * B[16,K]^T through the same tile
 * primitive every GEMM below uses.  K multiple of 16. */
int pf_selftest_mfma(const float* a, const float* b, float* c, int K, pf_stream_t stream);
/* Cross-lane primitive self test (DPP / permlane swaps used by every reduction): in[64] -> out[10][64] =
 * lane^1, lane^2, lane^4, lane^8 values; v+v[lane^16]; v+v[lane^32]; 16-lane row sum; wave sum; wave max; row max. */
int pf_selftest_lanes(const float* in, float* out, pf_stream_t stream);

/* ---- fused row-linear -------------------------------------------------------------------
 * y = epilogue(x W^T + bias), replaces torch.nn.Linear / ipa_pytorch.Linear (ipa_pytorch.py:116-181)
 * plus the element-wise ops the reference applies right after it:
 *   v = x W^T + bias ; relu ; v *= row_mask (mask_pre) ; v += residual ; LayerNorm(gamma,beta) ;
 *   v *= row_mask (mask_post)
 * covering ga.py:94-95,103-111, ipa_pytorch.py:196-206,478-482, TransformerEncoderLayer's
 * projections.  K must be a multiple of 16 (host pads); LayerNorm needs N <= 128. */
typedef struct {
    const float* x;  int ldx;
    const float* w;  int ldw;       /* [N, ldw], first K columns used */
    const float* bias;              /* [N] or NULL */
    float* y;        int ldy;
    int M, N, K;
    int relu;
    const float* row_mask;          /* [M] or NULL */
    int mask_pre, mask_post;
    const float* residual; int ldr; /* [M, ldr] or NULL */
    const float* ln_gamma; const float* ln_beta; float ln_eps; /* NULL = no LayerNorm */
    const void* w_f16;              /* optional: W pre-split into fragment-order f16 hi/lo planes (engine.split_f16);
                                       selects the split-precision MFMA path (bias / ReLU / row mask / gate / residual,
                                       K % 32 == 0, no LayerNorm) */
    const float* gate; int ldg;     /* split path only, optional: y = gate[m,n] > 0 ? y : 0 before the residual is added
                                       (backward of a ReLU fused into the dx product of the training path) */
    /* split path only, optional (IPA projection of the inference plan): output features n >= pt_col0 are point coordinates
     * packed (x, y, z, 0) per point -- 64 query points (h*8+p) then 160 key/value points (h*20+p; p < 8 key, else value) --
     * and are not stored to y: each point goes through the residue frame, R p + t (rigid_utils.py:1124-1150 via
     * ipa_pytorch.py:360-388), straight into qp [M,192] / kp [M,192] / vp [M,288] (what pf_ipa_points_fwd produces). */
    const float* pt_rot; const float* pt_trans; float* pt_qp; float* pt_kp; float* pt_vp; int pt_col0;
    int single_pass;                /* split path, 4-wave form only: 1 = f16 precision mode (one f16 MFMA per product, hi planes only) */
    /* split path with pt_* set (IPA projection of the inference plan), optional: write the attention operands as f16 planes
     * instead of fp32 `y` / `pt_vp` (csrc/ipa_split.hip consumes them; L = att_L must be a multiple of 16, M = B L):
     * single_pass only since ABI 50 (the hi / lo plane form of the fp32 mode was removed; att_qk without single_pass is refused):
     *   att_qk : M x 2048 f16 in all: the q rows [M][1024], then the k rows in the fragment order of the score kernel's first product
     *            (ABI 53) -- block (sample, head, 16-key tile, 32-channel K-step) = 512 f16 = the eight operand slots of each of the 64
     *            lanes: key tile row r, channel group kg = (c / 8) % 4 at lane kg * 16 + r, slot c % 8 (natural K order)
     *   att_vt : values TRANSPOSED per (sample, head), [B][8][PF_ATT_VT_HEAD(L)] f16 in the fragment order described at PF_ATT_VROWS
     *            (ABI 53; until then [PF_ATT_VROWS][L] rows: an operand load of the score kernel touched sixteen rows, 64 bytes of each) */
    void* att_qk; void* att_vt; int att_L;
    /* optional (split path): the rows are [B][key_L] residues and key_end[b] (device memory, int32 [B]) = 1 + the last unmasked
     * residue of sample b (the same list as pf_ipa_attn_args.key_end).  A row tile that lies entirely at or beyond its samples'
     * key ends is SKIPPED: x is not read, y / the point and attention outputs of those rows are NOT written (their consumers skip
     * the same rows; the caller keeps the buffers finite).  active_rows (host side hint, 0 = unknown) = sum of key_end: lets the
     * launcher pick the rows-persistent kernel when the ACTIVE tiles fit whole rounds of workgroups. */
    const int* key_end; int key_L; int active_rows;
    /* optional (split path with pt_* set, fp32 output, att_L = L a multiple of 16, M = B L; ABI 54): the k columns of the packed IPA
     * projection (features 1024 + 256 h .. + 127) go to k_frag INSTEAD of y, as fp32 fragments of the score kernel's first product:
     * block (sample, head, 16-key tile, 16-channel step) = 256 floats = the float4 of each of the 64 lanes (key r, channels 4 g ..);
     * M x 1024 floats in all.  pf_ipa_attn_args.k_frag takes the same buffer. */
    float* k_frag;
} pf_linear_args;
int pf_linear_fwd(const pf_linear_args* a, pf_stream_t stream);
/* W [N,K] fp32 (ldw) -- or W^T when `transpose` (then w is [K,N], ldw >= N) -- to the fragment-order f16 hi/lo planes
 * pf_linear_args.w_f16 expects ([2][ceil16(N)/16][K/32][64][8] f16, N zero-padded to 16): the device-side form of
 * engine.split_f16, used per step by the training path (weights change every step).  Layout only. */
int pf_split_pack_f16(const float* w, int ldw, int N, int K, int transpose, void* out, pf_stream_t stream);
/* the same for many matrices in ONE launch: desc (DEVICE memory, ndesc entries, ascending `first`) describes each matrix as above
 * (N rows of the packed matrix, K % 32 == 0) and the index of its first work item (= 8 consecutive k of one output row:
 * Npad * K / 8 items per matrix); total_items = the sum.  Same planes as pf_split_pack_f16, optional range flag as in *_checked. */
typedef struct { const float* w; int ldw, N, K, transpose; void* out; int first; int pad_; } pf_pack_desc;
int pf_split_pack_f16_batch(const pf_pack_desc* desc_dev, int ndesc, int total_items, int* range_flag, pf_stream_t stream);
/* the same, and *range_flag (device int, zeroed by the caller) is set to 1 if any |w| exceeds the f16 range (65504) or is not
 * finite: the split representation cannot carry such a weight (the packed value saturates); callers treat it as an error */
int pf_split_pack_f16_checked(const float* w, int ldw, int N, int K, int transpose, void* out, int* range_flag, pf_stream_t stream);

/* ---- input mixing features: ga.py:94 (cat) + ga.py:79-85 / utils.py:60-71 (time embedding) +
 * layers.py:92-113 (AngularEncoding, 12 funcs) + nn.Embedding lookup.
 * out[B*L, 640] = [node_embed 128 | seq_emb[seqs] 128 | time 128 | angle code 245 | 0 x 11] */
typedef struct {
    const float* node_embed;   /* [B*L,128] */
    const float* seq_table;    /* [22,128] */
    const int64_t* seqs;       /* [B*L] */
    const float* t;            /* [B] */
    const float* time_freq;    /* [64] host-computed exp(-k ln(2056)/63) */
    const float* ang_freq;     /* [24] AngularEncoding.freq_bands */
    const float* angles;       /* [B*L,5] */
    float* out;                /* [B*L,640] */
    int B, L;
} pf_embed_args;
int pf_embed_inputs_fwd(const pf_embed_args* a, pf_stream_t stream);
/* the same features, the two Linears of res_feat_mixer (ga.py:94-96: 640 -> 128 ReLU -> 128, masked) and the quaternion of
 * the current frames (rigid_utils.py:208-227) in ONE launch (csrc/node_track.hip: input_mixer_kernel); w0_f16 / w2_f16 are
 * the fragment-order f16 planes of mixer.0 (K padded 629 -> 640) and mixer.2 */
typedef struct {
    const float* node_embed; const float* seq_table; const int64_t* seqs; const float* t; const float* time_freq;
    const float* ang_freq; const float* angles;                       /* as in pf_embed_args */
    const void* w0_f16; const float* b0; const void* w2_f16; const float* b2;
    const float* mask;         /* [B*L] */
    const float* rot;          /* [B*L,9] current frames */
    float* quat;               /* [B*L,4] */
    float* s_out;              /* [B*L,128] */
    int B, L;
    int single_pass;           /* 1 = f16 precision mode: one f16 MFMA per product (hi weight planes only) */
} pf_input_mixer_args;
int pf_input_mixer_fwd(const pf_input_mixer_args* a, pf_stream_t stream);

/* ---- rigid-body point projection: r.apply() on the IPA points, ipa_pytorch.py:360-387,
 * rigid_utils.py:1124 / 82-106.  proj row = [q|kv|q_pts|kv_pts]; outputs global-frame points. */
typedef struct {
    const float* proj; int ldp;    /* [B*L, ldp>=3744] */
    const float* rot;              /* [B*L,9] */
    const float* trans;            /* [B*L,3] */
    float* qp;                     /* [B*L, 8*8*3]  (h,p,xyz) */
    float* kp;                     /* [B*L, 8*8*3] */
    float* vp;                     /* [B*L, 8*12*3] */
    int rows;
} pf_ipa_points_args;
int pf_ipa_points_fwd(const pf_ipa_points_args* a, pf_stream_t stream);

/* ---- invariant point attention core: ipa_pytorch.py:389-475 (scores from scalar qk + pair bias +
 * point distances, masked softmax, o / o_pt / o_pair, inverse-frame projection, norms).
 * Reads z twice per query tile (bias pass, pair-value pass); never materialises the
 * [B,L,L,H,Pq,3] / [B,H,3,L,L,Pv] intermediates of the reference. */
typedef struct {
    const float* proj; int ldp;    /* [B*L, ldp] q at 0, kv at 1024 (per head: k 128 | v 128) */
    const float* qp; const float* kp; const float* vp; /* from pf_ipa_points_fwd */
    const float* z;                /* [B,L,L,64] */
    const float* rot; const float* trans; /* frames [B*L,9],[B*L,3] */
    const float* mask;             /* [B*L] */
    const float* w_b; const float* b_b;   /* linear_b  [8,64],[8]  */
    const float* w_dz; const float* b_dz; /* down_z    [16,64],[16] */
    const float* head_w;           /* [8] raw (softplus applied inside) */
    float* feats;                  /* [B*L,1536] */
    int B, L;
    /* optional: sqrt(1/3) (W_b z + b_b) precomputed per pair, [B,8,L,L] (head-major), by the producer of z
     * (pf_edge_transition_fwd bias_out, or pf_pair_bias_fwd): the bias pass over z is skipped, z is read once. */
    const float* bias;
    /* optional: the attention probabilities [B,8,L,L] are written here (the training backward keeps them).
     * With BOTH bias and p_out set and L <= 256 the two-kernel form runs (csrc/ipa_split.hip: scores per (sample, head),
     * then one streaming pass over z); otherwise the one-kernel form (csrc/ipa_attn.hip). */
    float* p_out;
    int variant;                   /* 0 = automatic; 1 = force the one-kernel form; 2 = demand the two-kernel form (error if impossible) */
    /* optional (two-kernel form, L % 16 == 0): the q / k / v operands as the f16 planes pf_linear_fwd (att_*) wrote; the score
     * kernel then runs its products on the f16 matrix instruction (single pass, att_mode = 2: the f16 mode) and `proj` / `vp` are
     * not read.  att_mode = 1 (hi / lo planes, 3-MFMA split) existed until ABI 50 as a test-only form and is refused now. */
    const void* att_qk; const void* att_vt; int att_mode;
    int head_group;                /* one-kernel form: 0 = head-group split by size; 2 / 4 / 8 = that variant */
    /* optional (two-kernel form): key_end[b] = 1 + the last unmasked residue of sample b (device memory, int32 [B]).  Keys and
     * query rows from key_end[b] on are skipped: their probabilities are exactly zero (mask term -1e5, ipa_pytorch.py:427-430) and
     * the outputs of masked query rows are multiplied by the mask afterwards (ga.py:104); feats / p_out there are NOT written. */
    const int* key_end;
    int z_f16;                     /* two-kernel form: z is [B,L,L,64] f16 (the f16 mode's pair tensor, see pf_edge_transition_args) */
    /* optional (two-kernel form): dz [B,L,L,16] fp32 = W_dz z without the bias (pf_edge_transition_args.dz_out, or pf_linear_fwd
     * of z for a pair tensor EdgeTransition did not produce): the pair aggregation sum_j P (W_dz z_j) + b_dz reads it instead of
     * z (which may then be NULL) */
    const float* dz;
    int dz_f16;                    /* dz is [B,L,L,16] f16 (the f16 mode: pf_edge_transition_args.dz_out_f16) */
    /* optional (two-kernel form with dz): 1 = the pair aggregation runs INSIDE the score kernel (the probabilities never leave the
     * workgroup: p_out is not written and may be NULL, no second kernel).  Needs fp32 dz with fp32 operands (proj) or f16 dz with
     * the f16 operand planes (att_*); other combinations fall back to the two-kernel form (which needs p_out). */
    int fused_pair;
    /* optional (two-kernel form, fp32 operands, L % 4 == 0, every query tile of a sample in one score workgroup: 64 <= L <= 128):
     * THE PROJECTION RUNS INSIDE THE SCORE KERNEL (ABI 50).  s_in [B*L,128] = the node state; proj_w_f16 = fragment-order hi / lo
     * planes (engine.split_f16) of the packed projection [3968,128] = [linear_q 1024 | linear_kv 2048 | 64 query points (x,y,z,0) |
     * 160 key / value points (x,y,z,0)] -- the matrix pf_linear_fwd takes with pt_col0 = 3072 -- and proj_bias [3968] its bias.
     * Each (sample, head) workgroup forms q and the query / key / value points (global frame, through rot / trans) on chip; `proj`
     * and qp / kp / vp are not touched in this form (proj must still be non-NULL).
     * fp32 operands (att_mode 0), ABI 51: `att_vt` is REQUIRED as the launch's SCRATCH of B x 8 x 512 x ceil32(L) f16 (1 KiB per key
     * and head; written through the const pointer): per (sample, head) the values as hi | lo f16 operand fragments of the second
     * product (transposed, fragment order: 1 KiB per (16-channel tile, 32-key step, plane)) and the k rows as operand fragments of
     * the first (8 KiB per 16-key tile: hi | lo f16 per 32-channel K-step without fused_pair, fp32 per 16-channel step with it) --
     * every operand load / store of the workgroup is one contiguous KiB.  The second product -- and without fused_pair the first --
     * runs as three f16 MFMAs per product (hi hi + hi lo + lo hi, ~2^-22 relative: the precision of every split-precision Linear of
     * this library) instead of fp32 MFMAs.  The buffer must hold FINITE values on entry (zero it
     * once): key columns at or beyond a sample's key end are not written and meet zero probabilities.
     * q, k and the points are bit-identical to pf_linear_fwd followed by the plain call; the outputs agree with it to ~1e-6 relative (ipa_pytorch.py:347-387 + 389-475 in one launch). */
    const float* s_in; const void* proj_w_f16; const float* proj_bias;
    /* optional (two-kernel form, fp32 operands without s_in, L % 16 == 0; ABI 54): the k rows as fp32 fragments written by pf_linear_fwd
     * (pf_linear_args.k_frag): the first product reads them instead of the k columns of `proj` -- one contiguous KiB per load instead of
     * sixteen rows x 64 bytes.  Same values, same arithmetic: results are bit-identical to the call without it. */
    const float* k_frag;
    /* optional (with s_in; ABI 58): 1 = THE KEYS ARE THE NODE STATE.  q_h . k_h = s_i^T (W_q,h^T W_k,h) s_j + terms constant along a softmax
     * row (ipa_pytorch.py:389-404,427-432), so when proj_w_f16 / proj_bias hold the query rows as W_k,h^T (W_q,h s + b_q,h)
     * (engine.fold_keys_into_queries: weights only) the k operand of the first product is the row of s_in itself: the kernel writes its
     * fragments from s_in and passes over the eight k tiles of the packed projection (whose rows are then never multiplied). */
    int k_from_s;
} pf_ipa_attn_args;
int pf_ipa_attn_fwd(const pf_ipa_attn_args* a, pf_stream_t stream);
/* 1 when pf_ipa_attn_fwd would run the projection-inside form (s_in) at this length (fp32 operands: f16_mode = 0; f16 operand
 * planes formed in LDS, att_mode 2: f16_mode = 1), 0 when it would refuse it (more than one score workgroup per (sample, head),
 * L % 4 / L % 16, LDS beyond 160 KiB): asked once per launch plan (ABI 55). */
int pf_ipa_proj_inside_ok(int L, int f16_mode);
/* the `bias` operand above for a pair tensor that EdgeTransition did not produce (block 0: edge_embed is constant over the
 * sampler steps -> computed once per sample() call): bias [B,8,L,L] = sqrt(1/3)(linear_b(z)), ipa_pytorch.py:393-404 */
int pf_pair_bias_fwd(const float* z, const float* w_b, const float* b_b, float* bias, int B, int L, pf_stream_t stream);

/* ---- sequence-transformer attention core (torch.nn.MultiheadAttention inside
 * nn.TransformerEncoderLayer, ga.py:53-62): 4 heads x 32, key padding mask. */
typedef struct {
    const float* qkv;              /* [B*L,384] q|k|v */
    const float* mask;             /* [B*L] */
    float* out;                    /* [B*L,128] */
    int B, L;
} pf_seq_attn_args;
int pf_seq_attn_fwd(const pf_seq_attn_args* a, pf_stream_t stream);

/* ---- fused node track of one GAEncoder block (csrc/node_track.hip) ----------------------------------
 * pf_node_head_fwd: s_ipa = LayerNorm(s + mask * linear_out(feats)) (ga.py:103-104, ipa_pytorch.py:478-482)
 *                   and qkv = in_proj(s_ipa) of seq_tfmr layer 0.  s_ipa may alias s_in. */
typedef struct {
    const float* feats;            /* [rows,1536] */
    const float* s_in;             /* [rows,128] */
    const float* mask;             /* [rows] */
    const void* w_out_f16; const float* b_out; /* ipa linear_out [128,1536], split planes */
    const float* ln_g; const float* ln_b;     /* ipa_ln */
    const void* w_in_f16; const float* b_in;   /* seq_tfmr layers.0.self_attn.in_proj [384,128], split planes */
    float* s_ipa;                  /* [rows,128] */
    float* qkv;                    /* [rows,384] */
    int rows;
    int single_pass;               /* 1 = f16 precision mode: one f16 MFMA per product (hi weight planes only) */
    /* optional: rows = [B][key_L]; row tiles entirely at or beyond their samples' key_end[b] are skipped (see pf_linear_args) */
    const int* key_end; int key_L;
    /* optional (training forward, fp32-parity mode): a0 = s_in + mask * linear_out(feats), the LayerNorm's input, [rows,128] */
    float* dump_a0;
    /* optional (ABI 57): 1 = the first 1024 columns of feats hold, per head h, that head's CONTRIBUTION to linear_out's output
     * (feats[:, 128 h + n] = sum_j P_h (W_out[:, 128 h : 128 h + 128] v_j)[n]: the value projection was multiplied by linear_out's
     * o-block when the weights were packed -- linear_out is linear and the softmax rows sum to one, ipa_pytorch.py:456,475-476).
     * The kernel then ADDS the eight head blocks (fixed order: ((h0 + h2) + h4) + h6, ((h1 + h3) + h5) + h7, the two sums) instead
     * of contracting them, and w_out_f16 is the [128, 512] matrix of the remaining columns (o_pt | o_pt_norm | o_pair): a third of
     * the weight stream and of the matrix work of this kernel.  Not with dump_a0. */
    int o_premul;
} pf_node_head_args;
int pf_node_head_fwd(const pf_node_head_args* a, pf_stream_t stream);

/* pf_node_tfmr_fwd: one post-LN nn.TransformerEncoderLayer (ga.py:53-62,105-106; d=128, 4 heads, ffn 128,
 * key padding mask) for 16 query rows per workgroup, followed by
 *   last == 0: the next layer's in_proj (qkv_out must not alias qkv: other workgroups still read K/V);
 *   last == 1: the block tail -- post_tfmr + residual (ga.py:107), StructureModuleTransition + mask
 *              (ga.py:108-109), BackboneUpdate + Rigid.compose_q_update_vec (ga.py:110-113) and, if has_et,
 *              EdgeTransition.initial_embed + its per-residue terms pre[rows,512] (ipa_pytorch.py:234-243).
 * s_out may alias s_ipa; quat/rot/trans may be updated in place. */
typedef struct {
    const float* qkv;              /* [B*L,384] this layer's q|k|v */
    const float* resid;            /* [B*L,128] layer input */
    const float* mask;             /* [B*L] */
    /* every *_f16 weight is the reference matrix [N,K] pre-split into fragment-order f16 planes (engine.split_f16) */
    const void* w_o_f16; const float* b_o; const float* n1_g; const float* n1_b;
    const void* w_1_f16; const float* b_1; const void* w_2_f16; const float* b_2; const float* n2_g; const float* n2_b;
    const void* w_in_next_f16; const float* b_in_next; float* qkv_out; float* v_out;   /* last == 0 */
    int last;
    const float* s_ipa;                                                                /* last == 1 ... */
    const void* w_post_f16; const float* b_post;
    const void* w_t1_f16; const float* b_t1; const void* w_t2_f16; const float* b_t2; const void* w_t3_f16; const float* b_t3;
    const float* nt_g; const float* nt_b;
    const void* w_bb_f16; const float* b_bb;   /* b_bb: bb_update bias [6] zero-padded to 8 floats */
    float* s_out;
    const float* quat_in; const float* rot_in; const float* trans_in;
    float* quat_out; float* rot_out; float* trans_out;
    int has_et;
    const void* w_init_f16; const float* b_init; const void* w_pre_f16; const float* b_pre; float* pre;
    int B, L;
    /* optional (last == 1, has_et == 0, i.e. the final block): the two output heads ga.py:123-124 on the new node
     * state while it is in LDS -- seq_net / angle_net = Linear(128,128)+ReLU, Linear(128,128)+ReLU, Linear(128,20|5);
     * h_w[net][layer] are fragment-order f16 planes (last layers padded to 32 / 16 rows), h_b[net][layer] the biases. */
    const void* h_w[2][3]; const float* h_b[2][3];
    float* logits_out;             /* [B*L,20] */
    float* ang_out;                /* [B*L,5] (before the % 2pi of ga.py:125) */
    int single_pass;               /* 1 = f16 precision mode: one f16 MFMA per product (hi weight planes only); the attention core,
                                      LayerNorms, residuals and the frame update stay fp32 */
    /* optional: query-row tiles that start at or beyond key_end[b] of their sample are skipped -- nothing of theirs is written
     * (s_out, frames, pre, qkv_out, v_out, logits_out, ang_out keep what they held; see pf_linear_args.key_end) */
    const int* key_end;
    /* optional (training forward; fp32-parity mode, <= 256 row tiles of 16): dump[k] != NULL for k < 5 (last == 0) / k < 11
     * (last == 1) -> the intermediates the backward needs are also stored, fp32 [B*L,128] each:
     *   0 att (attention output before out_proj)   1 h = out_proj(att) + x   2 x1 = LN1(h)   3 f = relu(linear1(x1))
     *   4 h2 = linear2(f) + x1      (the layer output LN2(h2) is v_out for last == 0, dump 5 for last == 1)
     *   5 tf = LN2(h2)   6 s2 = s_ipa + post_tfmr(tf)   7 t1   8 t2 (StructureModuleTransition hidden)   9 h3 = linear_3(t2) + s2
     *   10 the backbone update [B*L,8] (6 used: BackboneUpdate output, input of compose_q_update_vec) */
    float* dump[11];
    /* optional (last == 1, L % 16 == 0): row_on[(b L + i) / 16] != 0 marks the 16-row groups whose outputs are wanted; a query tile
     * without a marked group is skipped (nothing of it is written).  The sampler marks the groups that hold a generated residue in
     * the LAST block of a step: the predictions of context residues are replaced by the context anyway (flow_model.py:291-311). */
    const int* row_on;
} pf_node_tfmr_args;
int pf_node_tfmr_fwd(const pf_node_tfmr_args* a, pf_stream_t stream);

/* ---- rot -> quat: rigid_utils.py:208-227 (top eigenvector of the 4x4 K matrix; the reference
 * calls torch.linalg.eigh, here a shifted power iteration converged to fp32). */
int pf_rot_to_quat(const float* rot, float* quat, int n, pf_stream_t stream);

/* ---- backbone update: Rigid.compose_q_update_vec, rigid_utils.py:1039-1063,587-616,266-275,
 * 331-332 and quat_to_rot 185-205.  In-place allowed. */
typedef struct {
    const float* quat_in;          /* [n,4] */
    const float* rot_in;           /* [n,9] rotation applied to the translation update */
    const float* trans_in;         /* [n,3] */
    const float* upd; int ldu;     /* [n,ldu>=6] */
    const float* mask;             /* [n] */
    float* quat_out; float* rot_out; float* trans_out;
    int n;
} pf_rigid_update_args;
int pf_rigid_update_fwd(const pf_rigid_update_args* a, pf_stream_t stream);

/* ---- EdgeTransition: ipa_pytorch.py:233-248 + edge mask ga.py:118.  The 192-wide concat
 * [z, n_i, n_j] is never built: pre[B*L,512] holds the per-residue terms
 *   a = W1[:,64:128] n, c = W1[:,128:192] n + b1, d = Wf[:,64:128] n, e = Wf[:,128:192] n + bf
 * (pf_node_tfmr_fwd tail, or pf_linear_fwd) and only the z part goes through the per-pair GEMMs.
 * Weights are passed PRE-SPLIT for the split-precision MFMA path: each *_f16 buffer holds the matrix as two
 * f16 planes (hi = f16(w), lo = f16((w - hi) * 2048)) in MFMA fragment order
 * [plane][N/16][K/32][64 lanes][8] -- produced by pepflowww_amd.engine.split_f16 (see csrc/common.h). */
typedef struct {
    const float* z_in;             /* [B*L*L,64] */
    float* z_out;                  /* may alias z_in; NULL (ABI 50, with bias_out set): z' is not stored -- the last EdgeTransition of a
                                    * step, whose output only feeds the next attention's pair bias / pair values (ga.py:115-118) */
    const float* pre;              /* [B*L,512] */
    const void* w1z_f16;           /* UNUSED since ABI 50 (operands of the round-1 tiled kernel, removed): leave NULL */
    const void* w2_f16;            /* UNUSED since ABI 50 */
    const float* b2;               /* trunk.2.bias [192] */
    const void* wf_f16;            /* UNUSED since ABI 50 */
    const float* ln_g; const float* ln_b;
    const float* mask;             /* [B*L] */
    int B, L;
    /* REQUIRED (this or w_stream32): trunk.0.weight[:, :64], trunk.2.weight and final_layer.weight as ONE 256 KiB stream of
     * 128 fragment pairs in consumption order (pepflowww_amd.engine.pack_et_stream) for the persistent LDS-ring kernel
     * (csrc/edge_transition_v3.hip).  The lo halves of the stream (and of
     * wb_frags) are f16(w - hi) UNSCALED -- this kernel adds all three products into one accumulator -- unlike the
     * w - hi times 2048 of every other split-precision operand (pf_split_pack_f16). */
    const void* w_stream;
    /* optional (persistent kernel only): also emit the NEXT block's IPA pair bias sqrt(1/3)(W_b z' + b_b)
     * from the normalised, masked z' while it is still in registers: bias_out [B,8,L,L] (head-major), wb_frags = linear_b
     * of the next block as 2 split-precision fragment pairs (pepflowww_amd.engine.pack_bias_frags), bb = its bias [8]. */
    float* bias_out; const void* wb_frags; const float* bb;
    /* optional (persistent kernel only; all three or none): the training forward keeps what the backward needs --
     * dump_h1 / dump_h2 [B*L*L,192] = relu(W1 x + b1), relu(W2 h1 + b2); dump_y [B*L*L,64] = the pre-LayerNorm output */
    float* dump_h1; float* dump_h2; float* dump_y;
    /* persistent kernel only: 1 = f16 precision mode -- ONE f16 MFMA per product (hi planes only) instead of the three of the
     * fp32-parity mode; fp32 accumulation, LayerNorm and storage unchanged.  Not combinable with the dumps. */
    int single_pass;
    /* optional (persistent kernel only): work list of the tiles that contain at least one unmasked pair.  A tile covers
     * pf_edge_transition_tile_rows(single_pass) rows i x 16 columns j of one sample; tile id = (b * nib + ib) * njb + jb with
     * nib = ceil(L / rows), njb = ceil(L / 16).  tile_list[0 .. *n_tiles) are processed, the others are NOT TOUCHED: z' of a fully
     * masked tile is exactly zero in the reference (ga.py:118), so the caller keeps those parts of z_out (and bias_out) zeroed.
     * Both pointers are device memory (the count is read by the kernel: no host synchronisation, graph-safe). */
    const int* tile_list; const int* n_tiles;
    /* f16 mode only (single_pass): the pair tensor stored as f16 -- z_in / z_out then point to [B*L*L,64] f16 (same element order).
     * z_out_f16 alone (block 0: fp32 edge embedding in, f16 out) or both. */
    int z_in_f16, z_out_f16;
    /* optional (persistent kernel, with bias_out): also emit the NEXT block's pair values W_dz z' (down_z, ipa_pytorch.py:440,
     * WITHOUT its bias) as dz_out [B*L*L,16] fp32, so that the next pf_ipa_attn_fwd (its `dz`) reads 64 bytes per pair instead
     * of the 256 of z'.  wb_frags then holds 6 KiB: the 2 fragment pairs of [linear_b (8 rows); down_z rows 0..7] followed by
     * rows 8..15 of down_z as 2 half fragments (pepflowww_amd.engine.pack_bias_frags(w_b, w_dz)).  Skipped tiles (tile_list)
     * are not touched: the caller keeps them zeroed. */
    float* dz_out;
    int dz_out_f16;                /* f16 mode only (single_pass): dz_out points to [B*L*L,16] f16 */
    /* optional: the same 256 KiB of weights packed for the 32x32x16 matrix instruction (pepflowww_amd.engine.pack_et_stream32:
     * 128 entries of [32 features x 16 K] hi | lo fragments in execution order) and the next block's [linear_b; down_z] tile in
     * that form (pack_bias_frags32, 8 KiB; needed with bias_out).  When w_stream32 is set and no dump_* is requested, the 32x32
     * kernel runs (csrc/edge_transition_v4.hip: tiles of 16 rows x 16 columns, pf_edge_transition_v4_tile_rows() for the work
     * list); everything else about the call is unchanged. */
    const void* w_stream32; const void* wb_frags32;
    /* optional (with the dumps): the ReLU gates [h1 > 0] / [h2 > 0] as one bit per (pair, feature), [B*L*L,24] bytes each: byte
     * 4 t + g of a pair holds the features 32 t + 16 h + 4 g + e at bit 4 h + e (t < 6, g < 4, h < 2, e < 4).  pf_et_bwd_chain
     * (m1 / m2) gates with them instead of reading the 768-byte activations back. */
    unsigned char* dump_m1; unsigned char* dump_m2;
    /* optional (32x32 kernel, fp32 pair tensor, L % 16 == 0; ABI 52): z_in / z_out in the kernel's FRAGMENT ORDER instead of
     * [B,L,L,64] -- the pair tensor between two EdgeTransition launches is read by nobody else (the attention takes bias_out / dz_out),
     * so the kernel may keep it in the order its lanes hold it: block ((b, 16 x 16 tile, wave w, 32-pair row pair) = 8 KiB) x piece k
     * (1 KiB) x lane (16 bytes), lane (rl, jl, g) = g * 32 + rl * 16 + jl of pair (i, j) = (16 ib + 2 w + rl, 16 jb + jl), piece k =
     * 4 mt + q holding channels 32 mt + 8 q + 4 g .. + 3.  Every load / store instruction of the pair tensor is then one contiguous
     * KiB (as [.., 64] rows an instruction touched 32 rows, 32 bytes of each).  w_stream32 must then have been packed with the
     * matching K order of the z operand (pepflowww_amd.engine.pack_et_stream32(..., z_frag=True)); pepflowww_amd.engine.z_to_frag /
     * z_from_frag convert a tensor.  Not with tile_list-skipped layouts other than whole tiles (the list is per tile anyway). */
    int z_in_frag, z_out_frag;
    /* optional (ABI 59; fp32-parity mode with z_in_frag / z_out_frag and bias_out, 32 <= L <= 4096): the same 256 KiB of weights in the
     * execution order of the hand-scheduled kernel (csrc/edge_transition_v5.hip: one 512-register wave per SIMD, every fragment
     * feeding 64 pairs, GEMM2 K-outer) -- pepflowww_amd.engine.pack_et_stream64(..., z_frag=True).  When set and the call has that
     * form, that kernel runs; any other form of call falls through to w_stream32 / w_stream as before.  Same inputs, outputs and
     * work-list semantics; results differ from the 32x32 kernel's only by the summation order inside the fp32 accumulators.
     * f16 mode (single_pass with z_in_f16 / z_out_f16 / dz_out_f16 and the frag flags): w_stream64 = the hi planes only (128 KiB,
     * pack_et_stream64(..., f16=True)) and the f16 pair tensor in THAT kernel's fragment order (pepflowww_amd.engine.z16_to_frag64: block
     * (tile, 32-pair group) of 4 KiB = K-step (1 KiB) x lane (the 8 halves of its MFMA operand)) -- not the 16x16x32 kernel's: a
     * single_pass call with w_stream64 set that the kernel does not cover is REFUSED (PF_E_BADARG) instead of falling through. */
    const void* w_stream64;
} pf_edge_transition_args;
int pf_edge_transition_fwd(const pf_edge_transition_args* a, pf_stream_t stream);
int pf_edge_transition_tile_rows(int single_pass);   /* rows i per tile of the persistent kernel (8; 16 in the f16 mode) */
int pf_edge_transition_v4_tile_rows(void);           /* ... of the 32x32 kernel (16) */

/* ---- encode(): once-per-call context featurisation (FlowModel.encode, flow_model.py:75-93) -------
 * node features: NodeEmbedder.forward up to the MLP input (node.py:35-99): aa embedding, per-aa-type
 * local heavy-atom coordinates, backbone dihedral code; plus the ground-truth frames
 * construct_3d_basis(CA, C, N) (geometry.py:89-111).  feat row = 1157 values padded to 1168;
 * the 4-layer MLP (node.py:20-25,101-102) then runs through pf_linear_fwd. */
typedef struct {
    const int64_t* aa; const int64_t* res_nb; const int64_t* chain_nb;  /* [B*L] */
    const float* pos;          /* [B*L,15,3] */
    const float* mask_atoms;   /* [B*L,15] 0/1 */
    const float* gen_mask;     /* [B*L] */
    const float* aa_table;     /* node_embedder.aatype_embed.weight [22,128] */
    const float* freq3;        /* dihed_embed.freq_bands [6] */
    float* feat;               /* [B*L,1168] */
    float* rot1; float* trans1;/* [B*L,9], [B*L,3] ground-truth frames */
    float* mres; float* ctx;   /* [B*L] residue mask (CA present), context mask (CA present & !generate) */
    int B, L;
    int sample_structure, sample_sequence;   /* cfg.interpolant flags (flow_model.py:86-87) */
} pf_node_feat_args;
int pf_node_features_fwd(const pf_node_feat_args* a, pf_stream_t stream);

/* edge features + all five Linears of EdgeEmbedder.forward (edge.py:39-111) fused; 64 pairs per
 * workgroup, the [B,L,L,225] distance tensor never leaves LDS.  w_d0 is distance_embed.0.weight
 * padded to [64,240], w_o0 is out_mlp.0.weight padded to [64,224] (MFMA K granularity). */
typedef struct {
    const int64_t* aa; const int64_t* res_nb; const int64_t* chain_nb;
    const float* pos; const float* mask_atoms;
    const float* ctx; const float* mres;           /* from pf_node_features_fwd */
    const float* aapair_table;  /* [484,64] */
    const float* relpos_table;  /* [65,64] */
    const float* distcoef;      /* aapair_to_distcoef.weight [484,225] (softplus applied inside) */
    const float* freq3;         /* dihedral_embed.freq_bands [6] */
    const float* w_d0; const float* b_d0; const float* w_d2; const float* b_d2;
    const float* w_o0; const float* b_o0; const float* w_o2; const float* b_o2; const float* w_o4; const float* b_o4;
    float* out;                 /* [B,L,L,64] */
    int B, L;
    int sample_structure, sample_sequence;
    /* optional (training path): intermediates the backward needs, per pair: Gaussian features [225], distance_embed.0 output
     * [64], the 224-wide concat tile, out_mlp.0 / .2 outputs [64].  dump_d2 is no longer written (ABI 48): pf_edge_distcoef_bwd
     * takes the squared distances from the features themselves; the field keeps the struct layout. */
    float* dump_g; float* dump_d2; float* dump_h1; float* dump_cat; float* dump_o1; float* dump_o2;
    /* optional workspace of 484*225 floats: when set, softplus(distcoef) is tabulated there first (one small launch) and the pair
     * kernel reads the table instead of evaluating log1p(exp(.)) per (pair, atom pair) -- same values */
    float* softplus_ws;
} pf_edge_feat_args;
int pf_edge_features_fwd(const pf_edge_feat_args* a, pf_stream_t stream);

/* ---- sampler state, flow_model.py:229-374 ------------------------------------------------
 * Device-resident sampler: the loop never syncs with the host.  `step` is a device counter so
 * that one captured hipGraph can be replayed for every step. */
typedef struct {
    /* ground truth / context (from encode) */
    const float* rot1; const float* trans1; const float* ang1; const int64_t* seq1;
    const float* gen_mask;         /* [B*L] 1 = generated residue */
    const float* res_mask;         /* [B*L] */
    /* current state (updated in place) */
    float* rot_t; float* trans_t; float* ang_t; int64_t* seq_t; float* simplex_t;
    /* initial noise kept for the Euler step (flow_model.py:318,328) */
    float* trans0; float* simplex0;
    /* network outputs of this step */
    const float* pred_rot; const float* pred_trans; const float* pred_ang_raw; const float* pred_logits;
    /* trajectory buffers [num_steps, ...]; slot `*step` is written */
    float* traj_rot; float* traj_trans; float* traj_ang; int64_t* traj_seq; float* traj_simplex;
    /* time grid ts[num_steps] (torch.linspace(1e-2,1,N), flow_model.py:280) */
    const float* ts; int num_steps;
    int* step;                     /* device int[2]: [0] step counter, [1] workgroup ticket of pf_sampler_step (zero between calls) */
    float* t_out;                  /* [B] time fed to the next network call */
    /* categorical noise: expo != NULL -> caller-supplied Exp(1) draws [2*num_steps,B*L,20];
     * NULL -> in-kernel Philox4x32-10 keyed by (seed, first_sample+b, draw, residue, class) */
    const float* expo; uint64_t seed; int64_t first_sample;
    int B, L;
    int sample_bb, sample_ang, sample_seq;
    /* optional: device uint64[2] = {seed, first_sample} read at run time (overrides the two fields above), so that the
     * hipGraph captured for one sample() call is replayed by the next call with another seed / shard offset */
    const uint64_t* seed_dev;
    /* optional (ABI 56): device int64 [B] = the GLOBAL index of each local sample (overrides first_sample + b).  A batch that was
     * re-ordered on the way in -- the length buckets of a ragged batch (pepflowww_amd/buckets.py) -- then draws exactly the streams
     * of the unpermuted run: the Philox key is (seed, sample_ids[b], draw, residue, class). */
    const int64_t* sample_ids;
} pf_sampler_args;
/* initial state from raw noise (flow_model.py:252-277): rot0/trans0_raw/ang0/simplex0_raw as drawn */
int pf_sampler_init(const pf_sampler_args* a, const float* rot0, const float* trans0_raw,
                    const float* ang0, const float* simplex0_raw, pf_stream_t stream);
/* post-process the prediction (291-312), record it, then Euler step (316-343) unless last step;
 * increments *step */
int pf_sampler_step(const pf_sampler_args* a, pf_stream_t stream);

/* stand-alone manifold steps (KAT surface): so3_utils.geodesic_t (500-520) and torus.tor_geodesic_t */
int pf_so3_geodesic(const float* base, const float* target, const float* t, float* out, int n, pf_stream_t stream);
int pf_so3_log(const float* rot, float* rotvec, int n, pf_stream_t stream);
int pf_so3_exp(const float* rotvec, float* rot, int n, pf_stream_t stream);
int pf_torus_geodesic(const float* base, const float* target, const float* t, float* out, int n, pf_stream_t stream);

/* ---- training forward, flow_model.py:111-227 (no autograd; the backward row is next) -------
 * corrupt: t = t_raw*(1-2*min_t)+min_t; generated residues are moved to time t on each manifold
 * (translations 131-134, SO(3) geodesic 136-138, torus geodesic 140-142, simplex + categorical
 * draw 149-155); context residues keep the clean values.
 * losses: trans / rot-vf / idealised-backbone / cross-entropy / angle-vf / torsion (161-227),
 * losses[6] in that order = mean over samples of per_sample[B,6]. */
typedef struct {
    /* clean data (from encode) */
    const float* rot1; const float* trans1; const float* ang1; const int64_t* seq1;
    const float* gen_mask; const float* res_mask;     /* [B*L] */
    /* raw noise as drawn: t_raw [B] U[0,1), rot0 [B*L,9], trans0_raw [B*L,3], ang0 [B*L,5], simplex0_raw [B*L,20] */
    const float* t_raw; const float* rot0; const float* trans0_raw; const float* ang0; const float* simplex0_raw;
    /* categorical noise: expo != NULL -> Exp(1) draws [2,B*L,20] (0: seq_t, 1: predicted seq); NULL -> Philox */
    const float* expo; uint64_t seed; int64_t first_sample;
    /* corrupted state (written by corrupt, read by losses) */
    float* t; float* rot_t; float* trans_t; float* ang_t; int64_t* seq_t;
    /* network prediction (losses only); pred_ang_raw = angle_net output before the % 2pi of ga.py:125 */
    const float* pred_rot; const float* pred_trans; const float* pred_ang_raw; const float* pred_logits;
    /* outputs of losses */
    int64_t* pred_seq;            /* [B*L] drawn sequence (optional, may be NULL) */
    float* per_sample;            /* [B,6] */
    float* losses;                /* [6] */
    int B, L;
    int sample_structure, sample_sequence;
    /* optional: the Philox seed read from DEVICE memory at run time (overrides `seed`), so that a training step
     * captured once as a hipGraph draws fresh categorical noise on every replay */
    const uint64_t* seed_dev;
} pf_train_args;
int pf_train_corrupt_fwd(const pf_train_args* a, pf_stream_t stream);
int pf_train_losses_fwd(const pf_train_args* a, pf_stream_t stream);
/* backward of sum_k w[k] * loss_k (train.py:121; weights learn_angle.yaml:37-43, order as losses[6]) with respect
 * to the four network outputs; `a` as passed to pf_train_losses_fwd (pred_seq must have been written by it).
 * First stage of the backward row: the gradients this call produces seed the trunk backward (not built yet). */
typedef struct {
    float w[6];
    float* d_rot;      /* [B*L,9]  d/d pred_rot   */
    float* d_trans;    /* [B*L,3]  d/d pred_trans */
    float* d_ang;      /* [B*L,5]  d/d pred_ang_raw (the % 2pi of ga.py:125 has unit slope) */
    float* d_logits;   /* [B*L,20] d/d pred_logits */
    const float* w_dev; /* optional: the six weights read from device memory (overrides w; graph-captured steps) */
} pf_train_bwd_args;
int pf_train_losses_bwd(const pf_train_args* a, const pf_train_bwd_args* g, pf_stream_t stream);

/* ---- building blocks of the training backward (train.py:133; csrc/backward.hip) -----------------------------
 * Correctness-first fp32 kernels from which the trunk backward is assembled (first users: output heads, final
 * backbone update).  A Linear y = x W^T + b (ipa_pytorch.Linear / nn.Linear) needs three GEMMs:
 *   forward y = x W^T : A = x (sam = ldx, sak = 1), B(k,n) = W[n,k] (sbk = 1, sbn = ldw)
 *   dx = dy W         : A = dy,                     B(k,n) = W[k,n] (sbk = ldw, sbn = 1)      [K = N_out]
 *   dW (+)= dy^T x    : A(m,k) = dy[k,m] (sam = 1, sak = ldy), B = x (sbk = ldx, sbn = 1)     [K = rows] */
typedef struct {
    const float* A; long long sam, sak;
    const float* B; long long sbk, sbn;
    float* C; int ldc;
    int M, N, K;
    int accumulate;                /* C += instead of C = */
    /* optional forward epilogue: C = relu?(alpha A B + bias[n]) + residual[m,n] (residual with C's leading dimension) */
    const float* bias; int relu; const float* residual;
    float alpha;                   /* scale of the product (set 1) */
    /* optional two-level batching (e.g. sample x head): slice (z1, z2) uses A + z1*bsA1 + z2*bsA2 etc.; 0 = no batching */
    int batch1, batch2;
    long long bsA1, bsA2, bsB1, bsB2, bsC1, bsC2;
    int ksplit;                    /* internal (set by the launcher): split-K factor for long-K, few-tile products */
    /* optional (no batching): rowsum_a[m] += sum_k A(m,k), added atomically -- the bias gradient db = column sums of dy falls
     * out of the dW = dy^T x product (A = dy^T) without a pass of its own; must hold zeros or a running sum */
    float* rowsum_a;
    /* optional: C = gate[m,n] > 0 ? C : 0 before the residual is added (gate with C's leading dimension): the ReLU backward of
     * the layer below, fused into dx = dy W */
    const float* gate;
} pf_gemm_args;
int pf_gemm_f32(const pf_gemm_args* a, pf_stream_t stream);
/* the two products of a row-sized Linear backward in ONE launch: dx = dy W (a1) and dW (+)= dy^T x (a2), each described as for
 * pf_gemm_f32.  Their workgroups share a grid (each product alone fills a fraction of the CUs for a chain of global round trips);
 * results are those of two pf_gemm_f32 calls bit for bit, and any pair outside the compiled-in layouts runs as exactly that. */
int pf_gemm_f32_dual(const pf_gemm_args* a1, const pf_gemm_args* a2, pf_stream_t stream);
/* n <= PF_GEMM_GROUP_MAX INDEPENDENT products (no output of one is an operand or the output of another), each described as for
 * pf_gemm_f32, in ONE launch: the batched sample x head products of the IPA backward (ipa_pytorch.py:389-475 reversed: g_q, g_k, g_v
 * and the three point contractions) are short latency chains whose workgroups overlap in a shared grid.  Results are those of n
 * pf_gemm_f32 calls bit for bit; products outside the compiled-in layouts / tile size run as exactly that. */
#define PF_GEMM_GROUP_MAX 6
int pf_gemm_f32_group(const pf_gemm_args* a, int n, pf_stream_t stream);
/* EdgeTransition operands of the TRAINING forward from the fp32 parameters, two launches (the parameters change every step):
 * stream_out = the persistent kernel's 256 KiB fragment stream (pf_edge_transition_args.w_stream: [128][hi 512 | lo 512] f16) as a
 * gather through stream_idx (int32 [128*512] flat indices into trunk.0.weight | trunk.2.weight | final_layer.weight) with
 * lo = f16((w - hi) * lo_scale); pre_w [512,64], pre_b [512] = weight / bias of the per-residue terms a | c | d | e. */
int pf_et_pack_train(const float* w1, const float* b1, const float* w2, const float* wf, const float* bf, const int* stream_idx,
                     void* stream_out, float lo_scale, float* pre_w, float* pre_b, pf_stream_t stream);

/* ---- the dx chain of the EdgeTransition backward in one kernel (csrc/et_bwd.hip; ipa_pytorch.py:233-248 reversed):
 *   g_u = g_y Wf;  g_h2 = g_u * [h2 > 0];  g_h1 = (g_h2 W2) * [h1 > 0];  g_x = g_h1 W1 + g_u
 * g_y [npairs,64] = gradient w.r.t. the pre-LayerNorm output; h1, h2 [npairs,192] = the saved hidden activations
 * (pf_edge_transition_args.dump_h1 / dump_h2); w*T_f16 = the TRANSPOSED weight matrices as fragment-order f16 hi / lo planes
 * (pf_split_pack_f16 with transpose = 1: final_layer [64,192] -> [192,64]; trunk.2 and trunk.0 [192,192]).  Outputs [npairs,192]:
 * g_h2 and g_h1 (already gated: the operands of the weight-gradient products), g_x (for pf_et_concat_bwd).  Same split-precision
 * arithmetic as three pf_linear_fwd products. */
typedef struct {
    const float* g_y; const float* h1; const float* h2;
    const void* wfT_f16; const void* w2T_f16; const void* w1T_f16;
    float* g_h2; float* g_h1; float* g_x;
    long long npairs;
    /* optional: the ReLU gates [h1 > 0], [h2 > 0] as bits, [npairs,24] bytes each (pf_edge_transition_args.dump_m1 / dump_m2); when both
     * are set the kernel reads them instead of h1 / h2 (which may then be NULL): 48 bytes per pair instead of 1536 */
    const unsigned char* m1; const unsigned char* m2;
} pf_et_bwd_args;
int pf_et_bwd_chain(const pf_et_bwd_args* a, pf_stream_t stream);

/* weight gradient of a Linear over all pairs in one pass: C[M,N] (+)= A^T B with A = dy [R,M] (lda), B = x [R,N] (ldb),
 * M <= 192, N <= 256 (multiples of 4), and optionally colsum_a[M] (+)= column sums of A (the bias gradient).  One workgroup owns
 * the whole C for its row range, so A and B are read once (csrc/backward.hip: gemm_tn_wide_kernel; for N <= 192 the product
 * runs on the split-precision f16 MFMA with transposing LDS reads, gemm_tn_split_kernel).  workspace (optional, device memory
 * private to the stream, workspace_elems >= 256 * (M * N + M) floats): the workgroups store their partial sums there and a
 * second kernel adds them up; without it they accumulate with device-scope atomics (~55 us slower at 256 workgroups). */
int pf_gemm_tn_wide(const float* A, int lda, int M, const float* B, int ldb, int N, float* C, int ldc, long long R,
                    int accumulate, float* colsum_a, int colsum_accumulate, float* workspace, long long workspace_elems,
                    pf_stream_t stream);
/* C[M,N] (+)= A^T (B + B2): the same contraction over all R rows with the B operand given as the SUM of two tensors (same ldb), added
 * while a chunk is staged -- final_layer's weight gradient g_y^T (h2 + x) in one pass over g_y (M <= 64, N <= 192, R % 32 == 0). */
int pf_gemm_tn_sum2(const float* A, int lda, int M, const float* B, const float* B2, int ldb, int N, float* C, int ldc, long long R,
                    int accumulate, float* colsum_a, int colsum_accumulate, float* workspace, long long workspace_elems,
                    pf_stream_t stream);   /* colsum_a, workspace: optional, as in pf_gemm_tn_wide */
/* the same two contractions with EdgeTransition's concatenated input x = [z_ij | n_i | n_j] [B L L, 192] gathered from z [B L L, 64] and
 * the per-residue n [B L, 64] while it is staged (x never exists):  C[M,192] (+)= A^T x  (B2 == NULL, M <= 192: trunk.0.weight's
 * gradient, A = the gated g_h1) or  C[M,192] (+)= A^T (B2 + x)  (M <= 64: final_layer.weight's, A = g_y, B2 = h2); L >= 32. */
int pf_gemm_tn_cat(const float* A, int lda, int M, const float* B2, const float* z, const float* n, int B, int L, float* C, int ldc,
                   int accumulate, float* colsum_a, int colsum_accumulate, float* workspace, long long workspace_elems, pf_stream_t stream);
int pf_colsum_f32(const float* x, int ld, int M, int N, float* out, int accumulate, pf_stream_t stream);   /* bias grads */
int pf_relu_bwd(const float* y, float* dy, long long n, pf_stream_t stream);                                /* dy *= (y > 0) */
int pf_relu_gate(const float* y, const float* src, float* dst, long long n, pf_stream_t stream);            /* dst = y > 0 ? src : 0 */
int pf_add_out(const float* a, const float* b, float* dst, long long n, pf_stream_t stream);                 /* dst = a + b */
/* nn.LayerNorm backward over the last dimension (N <= 256, eps 1e-5): dx; dgamma[n] += sum_m dy xhat and dbeta[n] += sum_m dy
 * (optional, both or neither; accumulated atomically per workgroup, so they must hold zeros or a running sum);
 * dgamma_rows[m,n] = dy xhat (optional, the unreduced contributions). */
typedef struct { const float* x; const float* dy; const float* gamma; float* dx; float* dgamma_rows; int M, N;
                 float* dgamma; float* dbeta;
                 /* optional scratch private to the stream (>= (M / 64 + 1) * 2 N floats): pair-sized inputs then write per-workgroup
                  * partial sums that a second kernel adds up, instead of thousands of atomics per column */
                 float* workspace; long long workspace_elems;
                 /* optional: dy[m, :] is multiplied by row_scale[m] on the way in (y * mask in the forward: no masked copy of dy) */
                 const float* row_scale; } pf_layernorm_bwd_args;
int pf_layernorm_bwd(const pf_layernorm_bwd_args* a, pf_stream_t stream);
int pf_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, int M, int N, pf_stream_t stream);
int pf_row_mask(float* x, const float* mask, int M, int N, pf_stream_t stream);        /* x[m,:] *= mask[m] */
int pf_add_inplace(float* dst, const float* src, long long n, pf_stream_t stream);     /* dst += src */
/* backward of the attention core of the sequence transformer (pf_seq_attn_fwd; ga.py:53-62): qkv [B*L,384],
 * g_out [B*L,128] -> g_qkv [B*L,384]; stats = scratch [B*4*L*3] floats.  Probabilities are recomputed. */
int pf_seq_attn_bwd(const float* qkv, const float* mask, const float* g_out, float* g_qkv, float* stats, int B, int L,
                    pf_stream_t stream);
/* reverse of pf_rigid_update_fwd (Rigid.compose_q_update_vec + quat_to_rot, rigid_utils.py:1039-1063,185-205):
 * upstream gradients w.r.t. the NEW rotation matrix / quaternion (optional) / translation -> gradients w.r.t. the
 * update vector [n,6], the old quaternion, the old translation and (optional) the rotation used for the translation
 * update; rot_is_from_quat != 0 folds that last term into g_quat_in (blocks >= 1, where R_old = quat_to_rot(q)). */
typedef struct {
    const float* quat_in; const float* rot_in; const float* upd; int ldu; const float* mask;
    const float* g_rot_out; const float* g_quat_out; const float* g_trans_out;
    float* g_upd; float* g_quat_in; float* g_trans_in; float* g_rot_in;
    int rot_is_from_quat;
    int n;
} pf_rigid_update_bwd_args;
int pf_rigid_update_bwd(const pf_rigid_update_bwd_args* a, pf_stream_t stream);
/* nn.Embedding backward: table_grad[c, :dim] = sum of g[r, :dim] (row stride ldg) over rows with idx[r] == c */
int pf_embedding_bwd(const float* g, int ldg, const int64_t* idx, int rows, int ncls, int dim, float* table_grad, pf_stream_t stream);
/* encoder backward helpers (node.py / edge.py): per-pair indices and masks of EdgeEmbedder.forward (aa-pair class, clamped
 * relative position + 32, same-chain flag, structure-pair mask, residue-pair mask; aa_node = masked residue type per row),
 * embedding-table gradient with atomics and an optional per-row scale, a masked column-slice copy, and the gradient of
 * aapair_to_distcoef through exp(-softplus(w) d^2) (edge.py:83-89). */
int pf_edge_index(const int64_t* aa, const int64_t* res_nb, const int64_t* chain_nb, const float* ctx, const float* mres,
                  int sample_structure, int sample_sequence, int* aap, int* rel, float* same, float* sp, float* mp,
                  int64_t* aa_node, int B, int L, pf_stream_t stream);
int pf_embedding_bwd_atomic(const float* g, int ldg, const int* idx, const float* scale, long long rows, int dim, float* table_grad,
                            pf_stream_t stream);
int pf_slice_relu_mask(const float* src, int lds, int off, const float* ref, int ldr, int off_r, const float* rowscale, float* dst,
                       long long rows, int width, pf_stream_t stream);
/* table_grad [484,225] += d/d aapair_to_distcoef.weight (edge.py:83-89): g_g [pairs, ldg >= 225] = gradient of the Gaussian features,
 * gfeat [pairs,225] = the features themselves (pf_edge_feat_args.dump_g: g = exp(-softplus(w) d2) * mask), aap [pairs] pair-type rows
 * (pf_edge_index), w = the coefficient table, ratio_ws = workspace of 484*225 floats.  The squared distance is taken from the feature
 * itself (-d2 = ln(g) / softplus(w) where g != 0), so the forward does not store it. */
int pf_edge_distcoef_bwd(const float* g_g, int ldg, const float* gfeat, const int* aap, const float* w, float* ratio_ws, long long pairs,
                         float* table_grad, pf_stream_t stream);
/* EdgeTransition in unfused (saved-activation) form for the training path: x [B*L*L,192] = [z_ij | n_i | n_j]
 * (ipa_pytorch.py:236-243), emask [B*L*L] = m_i m_j (optional); and the reverse scatter g_z (+)= g_x[:, :64],
 * g_n [B*L,64] = sum_j g_x[(i,j),64:128] + sum_j g_x[(j,i),128:192]. */
int pf_et_concat(const float* z, const float* n, const float* mask, float* x, float* emask, int B, int L, pf_stream_t stream);
int pf_et_concat_bwd(const float* gx, float* gz, int accumulate_gz, float* gn, int B, int L, pf_stream_t stream);
/* g_quat (+)= (d quat_to_rot(q)/dq)^T g_rot  (rigid_utils.py:185-205): rotation gradients of a block's IPA -> its quaternion */
int pf_quat_to_rot_bwd(const float* quat, const float* g_rot, float* g_quat, int n, int accumulate, pf_stream_t stream);

/* ---- backward of the IPA core (ipa_pytorch.py:389-475; csrc/ipa_bwd.hip), correctness-first -------------------
 * Forward operands as for pf_ipa_attn_fwd; g_feats [B*L,1536] is the gradient w.r.t. its output.  Call order:
 *   pf_ipa_bwd_rows  -> P, gA [B,8,L,L], g_opt [rows,288] (global-frame gradient of o_pt), g_frame_rows [rows,12]
 *                       (d/d trans 3 | d/d rot 9 of the inverse-frame projection), g_gamma_rows [rows,8]
 *   pf_ipa_bwd_pairs -> g_bias [pairs,8], g_pz [pairs,16], g_z [pairs,64] (+= if accumulate_gz)
 *   host GEMMs (pf_gemm_f32, batched over sample x head): g_q = s_qk gA K, g_k = s_qk gA^T Q, g_v = P^T g_o into
 *                       g_proj[:, :3072]; g_qp = gA KP, g_kp = gA^T QP [rows,192], g_vp = P^T g_opt [rows,288]
 *   pf_ipa_bwd_points -> g_proj[:, 3072:3744] (raw point projections) and += g_frame_rows (point transforms)
 *   pf_ipa_headw_bwd  -> d/d head_weights [8] from the column sum of g_gamma_rows */
typedef struct {
    const float* proj; int ldp; const float* qp; const float* kp; const float* vp; const float* z;
    const float* rot; const float* trans; const float* mask;
    const float* w_b; const float* b_b; const float* w_dz; const float* b_dz; const float* head_w;
    const float* g_feats;
    float* P; float* gA; float* g_opt; float* g_frame_rows; float* g_gamma_rows;
    float* g_bias; float* g_pz; float* g_z; int accumulate_gz;
    const float* g_qp; const float* g_kp; const float* g_vp; float* g_proj;
    int B, L;
    /* optional (pf_ipa_bwd_pairs): [pairs,24] = g_bias (8) | g_pz (16) per pair in ONE tensor instead of g_bias / g_pz (which may
     * then be NULL): the parameter gradients of linear_b and down_z (ipa_pytorch.py:352,355) become one [24,64] product over z */
    float* g_bp;
} pf_ipa_bwd_args;
int pf_ipa_bwd_rows(const pf_ipa_bwd_args* a, pf_stream_t stream);
/* the same stage from SAVED probabilities (a->P written by pf_ipa_attn_fwd, p_out) and batched GEMMs issued by the host
 * (pepflowww_amd/backward.py):  pf_ipa_bwd_opt: a->g_opt holds o_pt in the global frame (= P vp) on entry and its gradient
 * on return, + g_frame_rows;  pf_ipa_bwd_pairterm: a->gA (holding g_P so far) += (W_dz^T g_o_pair) . z;
 * pf_ipa_bwd_softmax: a->gA = P (g_P - sum_j P g_P) in place, + g_gamma_rows. */
int pf_ipa_bwd_opt(const pf_ipa_bwd_args* a, pf_stream_t stream);
int pf_ipa_bwd_pairterm(const pf_ipa_bwd_args* a, pf_stream_t stream);
int pf_ipa_bwd_softmax(const pf_ipa_bwd_args* a, pf_stream_t stream);
int pf_ipa_bwd_pairs(const pf_ipa_bwd_args* a, pf_stream_t stream);
int pf_ipa_bwd_points(const pf_ipa_bwd_args* a, pf_stream_t stream);
int pf_ipa_headw_bwd(const float* g_gamma, const float* head_w, float* g_head_w, pf_stream_t stream);

/* ---- full-atom reconstruction of sampled residues: models_con/torsion.py:140-226 (+ get_heavyatom_mask 121-138 and
 * the context merge of sample.py:104-108).  Tables: the reference's idealised rigid groups (constants.py), 21 residue
 * types: tab_rot [21,8,9], tab_trans [21,8,3], tab_group [21,14] (int32), tab_pos [21,14,3], tab_mask [22,15] (uint8),
 * frame_group[5] = rigid-group index of psi, chi1..4.  Optional outputs may be NULL. */
typedef struct {
    const float* rot; const float* trans;   /* backbone frames [rows,9], [rows,3] */
    const float* angles;                    /* [rows,5] psi, chi1..4 */
    const int64_t* aa;                      /* [rows] residue types */
    const float* tab_rot; const float* tab_trans; const int* tab_group; const float* tab_pos; const unsigned char* tab_mask;
    int frame_group[5];
    float* pos14;                           /* [rows,14,3] */
    float* frames_rot; float* frames_trans; /* [rows,6,9], [rows,6,3]: backbone, psi, chi1..4 */
    const float* gen_mask; const float* ctx_pos15; float* pos15_merged;   /* where(generate, pad15(pos14), context) */
    unsigned char* mask15;                  /* [rows,15] heavy-atom mask of the residue type */
    int rows;
} pf_full_atom_args;
int pf_full_atom_fwd(const pf_full_atom_args* a, pf_stream_t stream);

/* ---- backbone-only reconstruction: reconstruct_backbone, pepflow/modules/common/geometry.py:446-489 (N, CA, C from the
 * frame and the residue type's idealised coordinates, O from the psi frame with psi = dihedral(N_i, CA_i, C_i, N_{i+1}),
 * 0 at C-termini per topology.py:5-25) and the merge of models_con/sample.py:77-82 (save_samples_bb).
 * tab_bb [21,3,3], tab_o [21,3]: constants.py:878-887.  pos4 / the merged outputs may be NULL (not both). */
typedef struct {
    const float* rot; const float* trans;   /* [B*L,9], [B*L,3] */
    const int64_t* aa; const int64_t* chain_nb; const int64_t* res_nb;   /* [B*L] */
    const unsigned char* mask;              /* [B*L] res_mask */
    const float* tab_bb; const float* tab_o;
    float* pos4;                            /* [B*L,4,3]  N, CA, C, O */
    const float* gen_mask; const float* ctx_pos15; const unsigned char* ctx_mask15;
    float* pos15_merged; unsigned char* mask15;   /* where(generate, pad15(pos4) / first-4 mask, context) */
    int B, L;
} pf_backbone_atoms_args;
int pf_backbone_atoms_fwd(const pf_backbone_atoms_args* a, pf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
