"""pepflowww_amd: MI355X-native (gfx950) implementation of PepFlow's flow-matching denoise path.

Public surface mirrors the reference (models_con/flow_model.py, models_con/ga.py):
    FlowModel(cfg.model) ; FlowModel.sample(batch, num_steps, sample_bb, sample_ang, sample_seq)
    FlowModel.ga_encoder(t, R_t, x_t, angles_t, seqs_t, node_embed, edge_embed, generate_mask, res_mask)
All arithmetic of the denoise step runs in libpepflow_hip.so (include/pepflow_hip.h).
"""
from .config import AttrDict, default_config  # noqa: F401
from .flow_model import FlowModel, PepflowRangeError  # noqa: F401
from .modules import GAEncoder  # noqa: F401

__all__ = ["FlowModel", "GAEncoder", "AttrDict", "default_config", "PepflowRangeError"]
