// pf_edge_transition_fwd -- EdgeTransition (ipa_pytorch.py:233-248) + edge mask (ga.py:118): entry point and dispatch.
//
//   x = [z_ij, n_i, n_j];  h1 = relu(W1 x + b1);  h2 = relu(W2 h1 + b2);
//   y = Wf (h2 + x) + bf;  z' = LayerNorm(y) * m_i m_j
//
// Three persistent kernels implement it (NOTES.md 3.2): edge_transition_v5.hip (hand-scheduled assembly stream, one
// 512-register wave per SIMD: the inference step's fp32-parity calls), edge_transition_v4.hip (v_mfma_f32_32x32x16_f16, no loader waves: the
// fp32-parity mode of the inference step) and edge_transition_v3.hip (16x16x32 with loader waves: the f16 mode and the training
// forward with its activation dumps).  Both take the weights as ONE fragment stream in consumption order (engine.pack_et_stream*)
// and the per-residue parts of W1 x / Wf x as pre[B*L,512] (produced by the node-track tail).  The round-1 tiled kernel that used
// to live here (64 pairs per workgroup, activations as f16 planes in LDS, 590-620 us at B=64 x 128 against 381) was removed in
// round 4: nothing ran it but its own tests.
#include "common.h"
#include "../../include/pepflow_hip.h"

int pf_edge_transition_v3_launch(const pf_edge_transition_args* a, hipStream_t stream);   // edge_transition_v3.hip
int pf_edge_transition_v4_launch(const pf_edge_transition_args* a, hipStream_t stream);   // edge_transition_v4.hip
int pf_edge_transition_v5_launch(const pf_edge_transition_args* a, hipStream_t stream);   // edge_transition_v5.hip

extern "C" int pf_edge_transition_fwd(const pf_edge_transition_args* a, pf_stream_t stream) {
    if (!a || !a->z_in || (!a->z_out && !a->bias_out) || !a->pre || !a->b2 || !a->ln_g || !a->ln_b || !a->mask || a->B <= 0 || a->L <= 0)
        return PF_E_BADARG;
    if (a->w_stream64) {                                         // hand-scheduled form: covers the inference step's calls (fp32 mode, fragment order)
        const int rc = pf_edge_transition_v5_launch(a, (hipStream_t)stream);
        // (f16 mode: that kernel's fragment order of the f16 pair tensor is its own -- a call it does not take must not fall through
        //  to the 16x16x32 kernel, which would read the tensor in ANOTHER order)
        if (rc != PF_E_BADARG || a->single_pass) return rc;
    }
    if (a->w_stream32 && !(a->dump_h1 || a->dump_h2 || a->dump_y)) return pf_edge_transition_v4_launch(a, (hipStream_t)stream);
    if (a->w_stream) return pf_edge_transition_v3_launch(a, (hipStream_t)stream);
    return PF_E_BADARG;                                          // a weight stream is required (w1z_f16 / w2_f16 / wf_f16 alone: no kernel)
}
