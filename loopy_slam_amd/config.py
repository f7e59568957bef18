"""YAML configuration with the reference's `inherit_from` recursive merge (src/config.py:10-56) and
`get_model` (src/config.py:60-74 -> src/conv_onet/config.py:4-22)."""
import os

import yaml

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _resolve(path):
    return path if os.path.isabs(path) or os.path.exists(path) else os.path.join(_ROOT, path)


def update_recursive(dst, src):
    for k, v in src.items():
        if k not in dst:
            dst[k] = dict()
        if isinstance(v, dict):
            update_recursive(dst[k], v)
        else:
            dst[k] = v


def load_config(path, default_path=None):
    with open(_resolve(path), 'r') as f:
        special = yaml.full_load(f)
    parent = special.get('inherit_from')
    if parent is not None:
        cfg = load_config(parent, default_path)
    elif default_path is not None:
        with open(_resolve(default_path), 'r') as f:
            cfg = yaml.full_load(f)
    else:
        cfg = dict()
    update_recursive(cfg, special)
    return cfg


def get_model(cfg, eng=None):
    from .slam import NICER
    return NICER(cfg, eng=eng)
