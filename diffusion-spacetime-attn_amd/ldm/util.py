"""Small helpers kept from the reference's ldm/util.py surface (instantiate_from_config & friends)."""
import importlib


def exists(x):
    return x is not None


def default(val, d):
    if val is not None:
        return val
    return d() if callable(d) else d


def get_obj_from_str(string, reload=False):
    module, cls = string.rsplit(".", 1)
    m = importlib.import_module(module)
    if reload:
        importlib.reload(m)
    return getattr(m, cls)


def instantiate_from_config(config):
    """{'target': 'pkg.mod.Class', 'params': {...}} -> object (dicts or attribute-style configs)."""
    get = config.get if hasattr(config, "get") else (lambda k, d=None: getattr(config, k, d))
    target = get("target")
    if target is None:
        raise KeyError("Expected key `target` to instantiate.")
    params = get("params") or {}
    return get_obj_from_str(target)(**dict(params))


def count_params(model):
    return sum(p.numel() for p in model.parameters())
