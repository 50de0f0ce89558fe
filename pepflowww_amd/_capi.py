"""ctypes binding of include/pepflow_hip.h (libpepflow_hip.so, hand-written gfx950 kernels).

This is the ONLY compute backend of the package: there is no CPU / PyTorch fallback.  If the
shared library is missing, or a tensor is not a contiguous fp32/int64 tensor on a ROCm device,
the call raises -- it never silently degrades.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# (PF_LIB_PATH: another build of the same library -- same-box A/B runs of kernel variants, tools/dev)
LIB_PATH = os.environ.get("PF_LIB_PATH") or os.path.join(_HERE, "lib", "libpepflow_hip.so")
ABI_VERSION = 59

_fp = C.c_void_p
_i = C.c_int


class LinearArgs(C.Structure):
    _fields_ = [("x", _fp), ("ldx", _i), ("w", _fp), ("ldw", _i), ("bias", _fp), ("y", _fp), ("ldy", _i),
                ("M", _i), ("N", _i), ("K", _i), ("relu", _i), ("row_mask", _fp), ("mask_pre", _i),
                ("mask_post", _i), ("residual", _fp), ("ldr", _i), ("ln_gamma", _fp), ("ln_beta", _fp),
                ("ln_eps", C.c_float), ("w_f16", _fp), ("gate", _fp), ("ldg", _i),
                ("pt_rot", _fp), ("pt_trans", _fp), ("pt_qp", _fp), ("pt_kp", _fp), ("pt_vp", _fp), ("pt_col0", _i),
                ("single_pass", _i), ("att_qk", _fp), ("att_vt", _fp), ("att_L", _i), ("key_end", _fp), ("key_L", _i), ("active_rows", _i),
                ("k_frag", _fp)]


class EmbedArgs(C.Structure):
    _fields_ = [("node_embed", _fp), ("seq_table", _fp), ("seqs", _fp), ("t", _fp), ("time_freq", _fp),
                ("ang_freq", _fp), ("angles", _fp), ("out", _fp), ("B", _i), ("L", _i)]


class IpaPointsArgs(C.Structure):
    _fields_ = [("proj", _fp), ("ldp", _i), ("rot", _fp), ("trans", _fp), ("qp", _fp), ("kp", _fp), ("vp", _fp),
                ("rows", _i)]


class IpaAttnArgs(C.Structure):
    _fields_ = [("proj", _fp), ("ldp", _i), ("qp", _fp), ("kp", _fp), ("vp", _fp), ("z", _fp), ("rot", _fp),
                ("trans", _fp), ("mask", _fp), ("w_b", _fp), ("b_b", _fp), ("w_dz", _fp), ("b_dz", _fp),
                ("head_w", _fp), ("feats", _fp), ("B", _i), ("L", _i), ("bias", _fp), ("p_out", _fp),
                ("variant", _i), ("att_qk", _fp), ("att_vt", _fp), ("att_mode", _i), ("head_group", _i), ("key_end", _fp), ("z_f16", _i), ("dz", _fp), ("dz_f16", _i), ("fused_pair", _i),
                ("s_in", _fp), ("proj_w_f16", _fp), ("proj_bias", _fp), ("k_frag", _fp), ("k_from_s", _i)]


class InputMixerArgs(C.Structure):
    _fields_ = [("node_embed", _fp), ("seq_table", _fp), ("seqs", _fp), ("t", _fp), ("time_freq", _fp), ("ang_freq", _fp),
                ("angles", _fp), ("w0_f16", _fp), ("b0", _fp), ("w2_f16", _fp), ("b2", _fp), ("mask", _fp), ("rot", _fp),
                ("quat", _fp), ("s_out", _fp), ("B", _i), ("L", _i), ("single_pass", _i)]


class SeqAttnArgs(C.Structure):
    _fields_ = [("qkv", _fp), ("mask", _fp), ("out", _fp), ("B", _i), ("L", _i)]


class RigidUpdateArgs(C.Structure):
    _fields_ = [("quat_in", _fp), ("rot_in", _fp), ("trans_in", _fp), ("upd", _fp), ("ldu", _i), ("mask", _fp),
                ("quat_out", _fp), ("rot_out", _fp), ("trans_out", _fp), ("n", _i)]


class EdgeTransitionArgs(C.Structure):
    _fields_ = [("z_in", _fp), ("z_out", _fp), ("pre", _fp), ("w1z_f16", _fp), ("w2_f16", _fp), ("b2", _fp), ("wf_f16", _fp),
                ("ln_g", _fp), ("ln_b", _fp), ("mask", _fp), ("B", _i), ("L", _i), ("w_stream", _fp),
                ("bias_out", _fp), ("wb_frags", _fp), ("bb", _fp), ("dump_h1", _fp), ("dump_h2", _fp), ("dump_y", _fp),
                ("single_pass", _i), ("tile_list", _fp), ("n_tiles", _fp), ("z_in_f16", _i), ("z_out_f16", _i), ("dz_out", _fp), ("dz_out_f16", _i),
                ("w_stream32", _fp), ("wb_frags32", _fp), ("dump_m1", _fp), ("dump_m2", _fp), ("z_in_frag", _i), ("z_out_frag", _i), ("w_stream64", _fp)]


class SamplerArgs(C.Structure):
    _fields_ = [("rot1", _fp), ("trans1", _fp), ("ang1", _fp), ("seq1", _fp), ("gen_mask", _fp), ("res_mask", _fp),
                ("rot_t", _fp), ("trans_t", _fp), ("ang_t", _fp), ("seq_t", _fp), ("simplex_t", _fp),
                ("trans0", _fp), ("simplex0", _fp),
                ("pred_rot", _fp), ("pred_trans", _fp), ("pred_ang_raw", _fp), ("pred_logits", _fp),
                ("traj_rot", _fp), ("traj_trans", _fp), ("traj_ang", _fp), ("traj_seq", _fp), ("traj_simplex", _fp),
                ("ts", _fp), ("num_steps", _i), ("step", _fp), ("t_out", _fp),
                ("expo", _fp), ("seed", C.c_uint64), ("first_sample", C.c_int64),
                ("B", _i), ("L", _i), ("sample_bb", _i), ("sample_ang", _i), ("sample_seq", _i), ("seed_dev", _fp), ("sample_ids", _fp)]


class TrainArgs(C.Structure):
    _fields_ = [("rot1", _fp), ("trans1", _fp), ("ang1", _fp), ("seq1", _fp), ("gen_mask", _fp), ("res_mask", _fp),
                ("t_raw", _fp), ("rot0", _fp), ("trans0_raw", _fp), ("ang0", _fp), ("simplex0_raw", _fp),
                ("expo", _fp), ("seed", C.c_uint64), ("first_sample", C.c_int64),
                ("t", _fp), ("rot_t", _fp), ("trans_t", _fp), ("ang_t", _fp), ("seq_t", _fp),
                ("pred_rot", _fp), ("pred_trans", _fp), ("pred_ang_raw", _fp), ("pred_logits", _fp),
                ("pred_seq", _fp), ("per_sample", _fp), ("losses", _fp),
                ("B", _i), ("L", _i), ("sample_structure", _i), ("sample_sequence", _i), ("seed_dev", _fp)]


class TrainBwdArgs(C.Structure):
    _fields_ = [("w", C.c_float * 6), ("d_rot", _fp), ("d_trans", _fp), ("d_ang", _fp), ("d_logits", _fp), ("w_dev", _fp)]


class GemmArgs(C.Structure):
    _fields_ = [("A", _fp), ("sam", C.c_longlong), ("sak", C.c_longlong), ("B", _fp), ("sbk", C.c_longlong), ("sbn", C.c_longlong),
                ("C", _fp), ("ldc", _i), ("M", _i), ("N", _i), ("K", _i), ("accumulate", _i),
                ("bias", _fp), ("relu", _i), ("residual", _fp), ("alpha", C.c_float), ("batch1", _i), ("batch2", _i),
                ("bsA1", C.c_longlong), ("bsA2", C.c_longlong), ("bsB1", C.c_longlong), ("bsB2", C.c_longlong),
                ("bsC1", C.c_longlong), ("bsC2", C.c_longlong), ("ksplit", _i), ("rowsum_a", _fp), ("gate", _fp)]


class LayerNormBwdArgs(C.Structure):
    _fields_ = [("x", _fp), ("dy", _fp), ("gamma", _fp), ("dx", _fp), ("dgamma_rows", _fp), ("M", _i), ("N", _i),
                ("dgamma", _fp), ("dbeta", _fp), ("workspace", _fp), ("workspace_elems", C.c_longlong), ("row_scale", _fp)]


class RigidUpdateBwdArgs(C.Structure):
    _fields_ = [("quat_in", _fp), ("rot_in", _fp), ("upd", _fp), ("ldu", _i), ("mask", _fp),
                ("g_rot_out", _fp), ("g_quat_out", _fp), ("g_trans_out", _fp),
                ("g_upd", _fp), ("g_quat_in", _fp), ("g_trans_in", _fp), ("g_rot_in", _fp), ("rot_is_from_quat", _i), ("n", _i)]


class IpaBwdArgs(C.Structure):
    _fields_ = [("proj", _fp), ("ldp", _i), ("qp", _fp), ("kp", _fp), ("vp", _fp), ("z", _fp), ("rot", _fp), ("trans", _fp), ("mask", _fp),
                ("w_b", _fp), ("b_b", _fp), ("w_dz", _fp), ("b_dz", _fp), ("head_w", _fp), ("g_feats", _fp),
                ("P", _fp), ("gA", _fp), ("g_opt", _fp), ("g_frame_rows", _fp), ("g_gamma_rows", _fp),
                ("g_bias", _fp), ("g_pz", _fp), ("g_z", _fp), ("accumulate_gz", _i),
                ("g_qp", _fp), ("g_kp", _fp), ("g_vp", _fp), ("g_proj", _fp), ("B", _i), ("L", _i), ("g_bp", _fp)]


class FullAtomArgs(C.Structure):
    _fields_ = [("rot", _fp), ("trans", _fp), ("angles", _fp), ("aa", _fp), ("tab_rot", _fp), ("tab_trans", _fp), ("tab_group", _fp),
                ("tab_pos", _fp), ("tab_mask", _fp), ("frame_group", _i * 5), ("pos14", _fp), ("frames_rot", _fp), ("frames_trans", _fp),
                ("gen_mask", _fp), ("ctx_pos15", _fp), ("pos15_merged", _fp), ("mask15", _fp), ("rows", _i)]


class BackboneAtomsArgs(C.Structure):
    _fields_ = [("rot", _fp), ("trans", _fp), ("aa", _fp), ("chain_nb", _fp), ("res_nb", _fp), ("mask", _fp), ("tab_bb", _fp),
                ("tab_o", _fp), ("pos4", _fp), ("gen_mask", _fp), ("ctx_pos15", _fp), ("ctx_mask15", _fp), ("pos15_merged", _fp),
                ("mask15", _fp), ("B", _i), ("L", _i)]


class NodeFeatArgs(C.Structure):
    _fields_ = [("aa", _fp), ("res_nb", _fp), ("chain_nb", _fp), ("pos", _fp), ("mask_atoms", _fp), ("gen_mask", _fp),
                ("aa_table", _fp), ("freq3", _fp), ("feat", _fp), ("rot1", _fp), ("trans1", _fp), ("mres", _fp),
                ("ctx", _fp), ("B", _i), ("L", _i), ("sample_structure", _i), ("sample_sequence", _i)]


class EdgeFeatArgs(C.Structure):
    _fields_ = [("aa", _fp), ("res_nb", _fp), ("chain_nb", _fp), ("pos", _fp), ("mask_atoms", _fp), ("ctx", _fp),
                ("mres", _fp), ("aapair_table", _fp), ("relpos_table", _fp), ("distcoef", _fp), ("freq3", _fp),
                ("w_d0", _fp), ("b_d0", _fp), ("w_d2", _fp), ("b_d2", _fp), ("w_o0", _fp), ("b_o0", _fp),
                ("w_o2", _fp), ("b_o2", _fp), ("w_o4", _fp), ("b_o4", _fp), ("out", _fp), ("B", _i), ("L", _i),
                ("sample_structure", _i), ("sample_sequence", _i),
                ("dump_g", _fp), ("dump_d2", _fp), ("dump_h1", _fp), ("dump_cat", _fp), ("dump_o1", _fp), ("dump_o2", _fp),
                ("softplus_ws", _fp)]


class NodeHeadArgs(C.Structure):
    _fields_ = [("feats", _fp), ("s_in", _fp), ("mask", _fp), ("w_out_f16", _fp), ("b_out", _fp), ("ln_g", _fp),
                ("ln_b", _fp), ("w_in_f16", _fp), ("b_in", _fp), ("s_ipa", _fp), ("qkv", _fp), ("rows", _i), ("single_pass", _i), ("key_end", _fp), ("key_L", _i), ("dump_a0", _fp), ("o_premul", _i)]


class NodeTfmrArgs(C.Structure):
    _fields_ = [("qkv", _fp), ("resid", _fp), ("mask", _fp),
                ("w_o_f16", _fp), ("b_o", _fp), ("n1_g", _fp), ("n1_b", _fp), ("w_1_f16", _fp), ("b_1", _fp),
                ("w_2_f16", _fp), ("b_2", _fp), ("n2_g", _fp), ("n2_b", _fp),
                ("w_in_next_f16", _fp), ("b_in_next", _fp), ("qkv_out", _fp), ("v_out", _fp), ("last", _i),
                ("s_ipa", _fp), ("w_post_f16", _fp), ("b_post", _fp), ("w_t1_f16", _fp), ("b_t1", _fp), ("w_t2_f16", _fp),
                ("b_t2", _fp), ("w_t3_f16", _fp), ("b_t3", _fp), ("nt_g", _fp), ("nt_b", _fp), ("w_bb_f16", _fp),
                ("b_bb", _fp), ("s_out", _fp), ("quat_in", _fp), ("rot_in", _fp), ("trans_in", _fp),
                ("quat_out", _fp), ("rot_out", _fp), ("trans_out", _fp), ("has_et", _i),
                ("w_init_f16", _fp), ("b_init", _fp), ("w_pre_f16", _fp), ("b_pre", _fp), ("pre", _fp), ("B", _i), ("L", _i),
                ("h_w", (_fp * 3) * 2), ("h_b", (_fp * 3) * 2), ("logits_out", _fp), ("ang_out", _fp), ("single_pass", _i), ("key_end", _fp), ("dump", _fp * 11), ("row_on", _fp)]


class EtBwdArgs(C.Structure):
    _fields_ = [("g_y", _fp), ("h1", _fp), ("h2", _fp), ("wfT_f16", _fp), ("w2T_f16", _fp), ("w1T_f16", _fp),
                ("g_h2", _fp), ("g_h1", _fp), ("g_x", _fp), ("npairs", C.c_longlong), ("m1", _fp), ("m2", _fp)]


_SIGNATURES = {
    "pf_et_bwd_chain": ([C.POINTER(EtBwdArgs), _fp], _i),
    "pf_gemm_tn_cat": ([_fp, _i, _i, _fp, _fp, _fp, _i, _i, _fp, _i, _i, _fp, _i, _fp, C.c_longlong, _fp], _i),
    "pf_gemm_tn_sum2": ([_fp, _i, _i, _fp, _fp, _i, _i, _fp, _i, C.c_longlong, _i, _fp, _i, _fp, C.c_longlong, _fp], _i),
    "pf_et_pack_train": ([_fp, _fp, _fp, _fp, _fp, _fp, _fp, C.c_float, _fp, _fp, _fp], _i),
    "pf_abi_version": ([], _i),
    "pf_selftest_mfma": ([_fp, _fp, _fp, _i, _fp], _i),
    "pf_selftest_lanes": ([_fp, _fp, _fp], _i),
    "pf_linear_fwd": ([C.POINTER(LinearArgs), _fp], _i),
    "pf_embed_inputs_fwd": ([C.POINTER(EmbedArgs), _fp], _i),
    "pf_input_mixer_fwd": ([C.POINTER(InputMixerArgs), _fp], _i),
    "pf_ipa_points_fwd": ([C.POINTER(IpaPointsArgs), _fp], _i),
    "pf_ipa_attn_fwd": ([C.POINTER(IpaAttnArgs), _fp], _i),
    "pf_ipa_proj_inside_ok": ([_i, _i], _i),
    "pf_pair_bias_fwd": ([_fp, _fp, _fp, _fp, _i, _i, _fp], _i),
    "pf_seq_attn_fwd": ([C.POINTER(SeqAttnArgs), _fp], _i),
    "pf_node_head_fwd": ([C.POINTER(NodeHeadArgs), _fp], _i),
    "pf_node_tfmr_fwd": ([C.POINTER(NodeTfmrArgs), _fp], _i),
    "pf_rot_to_quat": ([_fp, _fp, _i, _fp], _i),
    "pf_rigid_update_fwd": ([C.POINTER(RigidUpdateArgs), _fp], _i),
    "pf_edge_transition_fwd": ([C.POINTER(EdgeTransitionArgs), _fp], _i),
    "pf_edge_transition_tile_rows": ([_i], _i),
    "pf_edge_transition_v4_tile_rows": ([], _i),
    "pf_node_features_fwd": ([C.POINTER(NodeFeatArgs), _fp], _i),
    "pf_edge_features_fwd": ([C.POINTER(EdgeFeatArgs), _fp], _i),
    "pf_sampler_init": ([C.POINTER(SamplerArgs), _fp, _fp, _fp, _fp, _fp], _i),
    "pf_sampler_step": ([C.POINTER(SamplerArgs), _fp], _i),
    "pf_train_corrupt_fwd": ([C.POINTER(TrainArgs), _fp], _i),
    "pf_train_losses_fwd": ([C.POINTER(TrainArgs), _fp], _i),
    "pf_train_losses_bwd": ([C.POINTER(TrainArgs), C.POINTER(TrainBwdArgs), _fp], _i),
    "pf_gemm_f32": ([C.POINTER(GemmArgs), _fp], _i),
    "pf_gemm_f32_dual": ([C.POINTER(GemmArgs), C.POINTER(GemmArgs), _fp], _i),
    "pf_gemm_f32_group": ([C.POINTER(GemmArgs), C.c_int, _fp], _i),
    "pf_colsum_f32": ([_fp, _i, _i, _i, _fp, _i, _fp], _i),
    "pf_gemm_tn_wide": ([_fp, _i, _i, _fp, _i, _i, _fp, _i, C.c_longlong, _i, _fp, _i, _fp, C.c_longlong, _fp], _i),
    "pf_split_pack_f16": ([_fp, _i, _i, _i, _i, _fp, _fp], _i),
    "pf_relu_bwd": ([_fp, _fp, C.c_longlong, _fp], _i),
    "pf_relu_gate": ([_fp, _fp, _fp, C.c_longlong, _fp], _i),
    "pf_add_out": ([_fp, _fp, _fp, C.c_longlong, _fp], _i),
    "pf_layernorm_bwd": ([C.POINTER(LayerNormBwdArgs), _fp], _i),
    "pf_layernorm_fwd": ([_fp, _fp, _fp, _fp, _i, _i, _fp], _i),
    "pf_row_mask": ([_fp, _fp, _i, _i, _fp], _i),
    "pf_add_inplace": ([_fp, _fp, C.c_longlong, _fp], _i),
    "pf_seq_attn_bwd": ([_fp, _fp, _fp, _fp, _fp, _i, _i, _fp], _i),
    "pf_rigid_update_bwd": ([C.POINTER(RigidUpdateBwdArgs), _fp], _i),
    "pf_quat_to_rot_bwd": ([_fp, _fp, _fp, _i, _i, _fp], _i),
    "pf_embedding_bwd": ([_fp, _i, _fp, _i, _i, _i, _fp, _fp], _i),
    "pf_edge_index": ([_fp, _fp, _fp, _fp, _fp, _i, _i, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _fp], _i),
    "pf_embedding_bwd_atomic": ([_fp, _i, _fp, _fp, C.c_longlong, _i, _fp, _fp], _i),
    "pf_slice_relu_mask": ([_fp, _i, _i, _fp, _i, _i, _fp, _fp, C.c_longlong, _i, _fp], _i),
    "pf_edge_distcoef_bwd": ([_fp, _i, _fp, _fp, _fp, _fp, C.c_longlong, _fp, _fp], _i),
    "pf_et_concat": ([_fp, _fp, _fp, _fp, _fp, _i, _i, _fp], _i),
    "pf_et_concat_bwd": ([_fp, _fp, _i, _fp, _i, _i, _fp], _i),
    "pf_ipa_bwd_rows": ([C.POINTER(IpaBwdArgs), _fp], _i),
    "pf_ipa_bwd_pairs": ([C.POINTER(IpaBwdArgs), _fp], _i),
    "pf_ipa_bwd_opt": ([C.POINTER(IpaBwdArgs), _fp], _i),
    "pf_ipa_bwd_pairterm": ([C.POINTER(IpaBwdArgs), _fp], _i),
    "pf_ipa_bwd_softmax": ([C.POINTER(IpaBwdArgs), _fp], _i),
    "pf_ipa_bwd_points": ([C.POINTER(IpaBwdArgs), _fp], _i),
    "pf_ipa_headw_bwd": ([_fp, _fp, _fp, _fp], _i),
    "pf_split_pack_f16_checked": ([_fp, _i, _i, _i, _i, _fp, _fp, _fp], _i),
    "pf_split_pack_f16_batch": ([_fp, _i, _i, _fp, _fp], _i),
    "pf_full_atom_fwd": ([C.POINTER(FullAtomArgs), _fp], _i),
    "pf_backbone_atoms_fwd": ([C.POINTER(BackboneAtomsArgs), _fp], _i),
    "pf_so3_geodesic": ([_fp, _fp, _fp, _fp, _i, _fp], _i),
    "pf_so3_log": ([_fp, _fp, _i, _fp], _i),
    "pf_so3_exp": ([_fp, _fp, _i, _fp], _i),
    "pf_torus_geodesic": ([_fp, _fp, _fp, _fp, _i, _fp], _i),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


class PepflowHipError(RuntimeError):
    pass


def load():
    """dlopen libpepflow_hip.so (built in-tree by pepflowww_amd/build.py).  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PepflowHipError(
            f"{LIB_PATH} is missing: the HIP extension is the only compute path of pepflowww_amd "
            f"(no CPU fallback). Build it with `python -m pepflowww_amd.build` (hipcc, gfx950).")
    lib = C.CDLL(LIB_PATH)
    for name, (argtypes, restype) in _SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.argtypes = argtypes
        fn.restype = restype
    if lib.pf_abi_version() != ABI_VERSION:
        raise PepflowHipError(f"ABI mismatch: library {lib.pf_abi_version()} vs binding {ABI_VERSION}")
    _lib = lib
    return lib


CALLS = 0          # C-ABI calls checked so far (GraphedTrainStep reports how many one captured step makes)


def check(code, what):
    global CALLS
    CALLS += 1
    if code != 0:
        raise PepflowHipError(f"{what} failed with code {code} "
                              f"({'bad argument' if code == -1 else 'problem too large' if code == -2 else 'hipError_t'})")


def dptr(t, dtype=torch.float32, name="tensor"):
    """Device pointer of a contiguous tensor on a ROCm device; raises otherwise (no fallback)."""
    if t is None:
        return None
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor")
    if not t.is_cuda:
        raise PepflowHipError(f"{name}: tensor is on {t.device}; pepflowww_amd only runs on a ROCm GPU "
                              f"(the CPU restatement lives in oracle/ and is test infrastructure)")
    if t.dtype != dtype:
        raise PepflowHipError(f"{name}: dtype {t.dtype}, expected {dtype}")
    if not t.is_contiguous():
        raise PepflowHipError(f"{name}: tensor must be contiguous")
    return t.data_ptr()


class capture_guard:
    """Around a hipGraph capture: the cyclic garbage collector is held off (NOT run).  A collection that happens to run DURING
    capture can finalise device objects of earlier work (a dropped engine's captured graphs, events, pooled blocks); destroying
    those while a stream is capturing invalidates the capture and aborts the process (seen once in ~15 runs of the GPU suite, inside the
    training step's capture: "Fatal Python error: Aborted ... Garbage-collecting").  Holding the collector off is all that needs:
    round 4 also ran a full gc.collect() on entry, which cost 40 - 70 ms per capture once a process held a few engines -- 85 - 135 ms
    of every FlowModel.sample() call that captured (BENCH_r04 per_call: 19 - 23 % overhead)."""
    def __enter__(self):
        import gc
        self._was = gc.isenabled()
        gc.disable()
        return self

    def __exit__(self, *exc):
        import gc
        if self._was:
            gc.enable()
        return False


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream
