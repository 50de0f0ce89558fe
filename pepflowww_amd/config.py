"""Config container with the reference's attribute names (configs/learn_angle.yaml:1-34).

The reference reads an EasyDict (pepflow/utils/misc.py:110-114); any object exposing the same
attributes works (EasyDict, OmegaConf, this AttrDict)."""


class AttrDict(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            v = AttrDict(v)
        super().__setitem__(k, v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    __setattr__ = __setitem__


def default_config():
    """The `model:` section of configs/learn_angle.yaml."""
    return AttrDict({
        "encoder": {
            "node_embed_size": 128, "edge_embed_size": 64,
            "ipa": {"c_s": 128, "c_z": 64, "c_hidden": 128, "no_heads": 8, "no_qk_points": 8, "no_v_points": 12,
                    "seq_tfmr_num_heads": 4, "seq_tfmr_num_layers": 2, "num_blocks": 6, "stop_grad": False},
        },
        "interpolant": {
            "min_t": 1e-2, "t_normalization_clip": 0.9, "sample_sequence": True, "sample_structure": True,
            "rots": {"train_schedule": "linear", "sample_schedule": "exp", "exp_rate": 10},
            "trans": {"train_schedule": "linear", "sample_schedule": "linear", "sigma": 1.0},
            "seqs": {"num_classes": 20, "simplex_value": 5.0},
            "sampling": {"num_timesteps": 100}, "self_condition": False,
        },
    })
