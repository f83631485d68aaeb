"""``hydra.utils.instantiate`` when Hydra is installed; otherwise a minimal stand-in that covers how
the reference uses it on this path (``_target_`` + kwargs with ``_recursive_: false``, or an already
callable factory)."""
import importlib


def _locate(path: str):
    mod, _, name = path.rpartition(".")
    return getattr(importlib.import_module(mod), name)


def instantiate(cfg, *args, **kwargs):
    if isinstance(cfg, type(None)):
        return None
    if callable(cfg) and not hasattr(cfg, "keys"):
        return cfg(*args, **kwargs)
    try:
        import hydra.utils as hu          # noqa: WPS433
        from omegaconf import DictConfig   # noqa: WPS433
        if isinstance(cfg, DictConfig):
            return hu.instantiate(cfg, *args, **kwargs)
    except ImportError:
        pass
    if hasattr(cfg, "keys") and "_target_" in cfg:
        conf = {k: cfg[k] for k in cfg.keys() if k not in ("_target_", "_recursive_", "_convert_", "_partial_")}
        conf.update(kwargs)
        return _locate(cfg["_target_"])(*args, **conf)
    raise TypeError(f"cannot instantiate {type(cfg).__name__}: expected a DictConfig/dict with _target_ or a callable")
