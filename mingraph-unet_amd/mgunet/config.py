"""configs/*.yaml loader with the reference's keys (scripts/train_end_to_end.py:92-103) plus a
build-only `backend:` section.  No validation beyond what the reference does."""
from __future__ import annotations

import os

import yaml

DEFAULT_CONFIG_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs")


def load_config(config_path, config_name):
    with open(os.path.join(config_path, config_name), "r") as f:
        return yaml.safe_load(f)


def get_config_recursively(config, key_path, default=None):
    cur = config
    for k in key_path.split("."):
        if isinstance(cur, dict) and k in cur:
            cur = cur[k]
        else:
            return default
    return cur


def build_from_config(config_path=DEFAULT_CONFIG_DIR):
    """-> (UNet, GATNetwork, PatchGraphConstructor, model_cfg, train_cfg) wired as
    scripts/train_end_to_end.py:122-152 does, with node features = patch mean of decoder_feats[0]."""
    from .gat import GATNetwork
    from .patch_graph import PatchGraphConstructor
    from .unet import UNet
    mc = load_config(config_path, "model.yaml")
    tc = load_config(config_path, "training.yaml")
    u, g = mc["unet"], mc["gat"]
    unet = UNet(u["in_channels"], u["out_channels"], u["init_features"], u["depth"])
    node_dim = g.get("node_feature_dim") or u["init_features"]
    gat = GATNetwork(node_dim, g["hidden_dim"], g["output_dim"], g["num_heads"], 1, g["dropout"], g["alpha"])
    pgc = PatchGraphConstructor(mc["graph_construction"]["patch_size"])
    return unet, gat, pgc, mc, tc
