"""YAML configuration: a chain of files linked by `inherit_from`, merged key by key with the most specific file winning
(the reference's semantics, src/config.py:10-56), and `get_model` (src/config.py:60-74 -> src/conv_onet/config.py:4-22)."""
import os

import yaml

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _read(path):
    if not (os.path.isabs(path) or os.path.exists(path)):
        path = os.path.join(_ROOT, path)
    with open(path, 'r') as f:
        return yaml.full_load(f) or {}


def update_recursive(dst, src):
    """Deep merge of `src` into `dst` in place: nested mappings are merged, everything else is overwritten."""
    stack = [(dst, src)]
    while stack:
        into, frm = stack.pop()
        for key, val in frm.items():
            if isinstance(val, dict):
                if not isinstance(into.get(key), dict):
                    into[key] = {}
                stack.append((into[key], val))
            else:
                into[key] = val
    return dst


def load_config(path, default_path=None):
    """Follow `inherit_from` from `path` to the root of the chain (the root falls back on `default_path`), then merge from the
    most general file down to `path`."""
    chain, nxt = [], path
    while nxt is not None:
        doc = _read(nxt)
        chain.append(doc)
        nxt = doc.get('inherit_from')
    cfg = _read(default_path) if default_path is not None else {}
    for doc in reversed(chain):
        update_recursive(cfg, doc)
    return cfg


def get_model(cfg, eng=None):
    from .slam import NICER
    return NICER(cfg, eng=eng)
