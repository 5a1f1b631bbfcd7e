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


def load_config(static_renderer="gnt", overrides=None):
    """Compose the config surface of the reference (configs/pgdvs.yaml defaults list):
    _basic + model + static_renderer + engine.render_cfg, from pgdvs_amd/configs/*.yaml."""
    import pathlib
    import yaml

    root = pathlib.Path(__file__).resolve().parent / "configs"
    cfg = yaml.safe_load((root / "_basic.yaml").read_text())
    cfg["model"] = yaml.safe_load((root / "model" / "pgdvs_renderer.yaml").read_text())
    sr = yaml.safe_load((root / "static_renderer" / f"{static_renderer}.yaml").read_text()) or {"_target_": None}
    cfg["static_renderer"] = sr
    cfg["tracker"] = {}
    cfg["engine"] = yaml.safe_load((root / "engine" / "evaluator_pgdvs.yaml").read_text())
    cfg = AttrDict.wrap(cfg)
    for dotted, v in (overrides or {}).items():
        node = cfg
        keys = dotted.split(".")
        for k in keys[:-1]:
            node = node[k]
        node[keys[-1]] = v
    return cfg
