"""Minimal stand-ins for the two Hydra/OmegaConf pieces the renderer plugin API touches
(``hydra.utils.instantiate`` on ``_target_`` strings and attribute-style config access),
so the package works where hydra is not installed.  When hydra IS installed (inside
pgdvs.engines) real DictConfig objects work unchanged: only ``cfg.key`` attribute access
and ``_target_`` are used."""
import importlib


class AttrDict(dict):
    """dict with attribute access, nested (the subset of DictConfig the renderers use)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    @staticmethod
    def wrap(obj):
        if isinstance(obj, dict):
            return AttrDict({k: AttrDict.wrap(v) for k, v in obj.items()})
        if isinstance(obj, (list, tuple)):
            return type(obj)(AttrDict.wrap(v) for v in obj)
        return obj


def instantiate(cfg, *args, **kwargs):
    """hydra.utils.instantiate for ``{_target_: 'pkg.mod.Class', **kwargs}`` nodes
    (recursive into nested ``_target_`` nodes, like Hydra's default)."""
    target = cfg["_target_"] if isinstance(cfg, dict) else getattr(cfg, "_target_")
    mod_name, _, attr = target.rpartition(".")
    cls = getattr(importlib.import_module(mod_name), attr)
    items = cfg.items() if hasattr(cfg, "items") else []
    params = {}
    for k, v in items:
        if k == "_target_":
            continue
        if hasattr(v, "items") and "_target_" in v and v["_target_"] is not None:
            v = instantiate(v)
        params[k] = v
    params.update(kwargs)
    return cls(*args, **params)


def _resolve(node, root):
    """the `${a.b}` interpolations of the config files (OmegaConf syntax), whole-value references only"""
    if isinstance(node, dict):
        return {k: _resolve(v, root) for k, v in node.items()}
    if isinstance(node, list):
        return [_resolve(v, root) for v in node]
    if isinstance(node, str) and node.startswith("${") and node.endswith("}") and node.count("${") == 1:
        cur = root
        for k in node[2:-1].split("."):
            if not isinstance(cur, dict) or k not in cur:
                return node  # e.g. ${hydra.job.name}: left for Hydra
            cur = cur[k]
        return _resolve(cur, root)
    return node


def load_config(static_renderer=None, overrides=None, **groups):
    """Compose the config surface like Hydra does for the reference (configs/pgdvs.yaml): the defaults list of
    pgdvs_amd/configs/pgdvs.yaml (_basic + engine + model + static_renderer + tracker + dataset), group choices
    overridable by keyword (``static_renderer="geo"``, ``engine="visualizer_pgdvs"``, ``tracker="tapnet"``),
    then dotted ``overrides``; ``${...}`` references to config keys are resolved."""
    import pathlib
    import yaml

    root = pathlib.Path(__file__).resolve().parent / "configs"
    top = yaml.safe_load((root / "pgdvs.yaml").read_text())
    if static_renderer is not None:
        groups["static_renderer"] = static_renderer
    cfg = {}
    for entry in top["defaults"]:
        if entry == "_self_":
            continue
        if isinstance(entry, str):
            cfg.update(yaml.safe_load((root / f"{entry}.yaml").read_text()) or {})
            continue
        (group, choice), = entry.items()
        choice = groups.pop(group, choice)
        cfg[group] = yaml.safe_load((root / group / f"{choice}.yaml").read_text()) or {"_target_": None}
    assert not groups, f"unknown config groups {sorted(groups)}"
    cfg = AttrDict.wrap(_resolve(cfg, cfg))
    for dotted, v in (overrides or {}).items():
        node = cfg
        keys = dotted.split(".")
        for k in keys[:-1]:
            node = node[k]
        node[keys[-1]] = v
    return cfg
