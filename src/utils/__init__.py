"""Minimal stand-ins for the template helpers `src/train.py` imports (reference src/utils/__init__.py re-exports them from
utils.py / instantiators.py / logging_utils.py / pylogger.py).  They are control-plane glue without arithmetic (SURVEY.md
section 2 row 19: out of scope); only what the entry point needs to compose and run is here."""
import logging
import warnings
from typing import Any, Callable, Dict, List, Optional

from src.utils.distributed import BackpropType, concat_gather_all_gpu, gather_tensor, get_rank  # noqa: F401


class RankedLogger(logging.LoggerAdapter):
    """Logger that prefixes the rank and (rank_zero_only) stays silent elsewhere (pylogger.py:7-51)."""

    def __init__(self, name: str = __name__, rank_zero_only: bool = False, extra=None):
        super().__init__(logging.getLogger(name), extra)
        self.rank_zero_only = rank_zero_only

    def log(self, level, msg, *args, rank: Optional[int] = None, **kwargs):
        if not self.isEnabledFor(level):
            return
        r = get_rank()
        if self.rank_zero_only and r != 0:
            return
        if rank is None or rank == r:
            self.logger.log(level, f"[rank: {r}] {msg}", *args, **kwargs)


log = RankedLogger(__name__, rank_zero_only=True)


def extras(cfg) -> None:
    """utils.py:16-45 without the interactive tag prompt and the rich tree."""
    ex = cfg.get("extras") if hasattr(cfg, "get") else None
    if not ex:
        return
    if ex.get("ignore_warnings"):
        warnings.filterwarnings("ignore")
    if ex.get("enforce_tags") and not cfg.get("tags"):
        raise ValueError("no tags given: set `tags=[...]` (extras.enforce_tags)")
    if ex.get("print_config") and get_rank() == 0:
        import yaml
        print(yaml.safe_dump(_plain({k: v for k, v in cfg.items() if not k.startswith("_")}), sort_keys=False))


def _plain(x):
    if isinstance(x, dict):
        return {k: _plain(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_plain(v) for v in x]
    return x


def task_wrapper(task_func: Callable) -> Callable:
    """utils.py:125-177: log the exception, re-raise."""
    def wrap(cfg):
        try:
            return task_func(cfg)
        except Exception:
            log.exception("training failed")
            raise
    return wrap


def get_metric_value(metric_dict: Dict[str, Any], metric_name: Optional[str]) -> Optional[float]:
    """utils.py:180-200."""
    if not metric_name:
        return None
    if metric_name not in metric_dict:
        raise KeyError(f"metric '{metric_name}' not found among {sorted(metric_dict)}")
    return float(metric_dict[metric_name])


def _instantiate(node):
    try:
        import hydra
        return hydra.utils.instantiate(node)
    except ImportError:
        from medmoe_amd.hydra_lite import instantiate
        return instantiate(node)


def instantiate_callbacks(callbacks_cfg) -> List[Any]:
    """instantiators.py:13-33: one object per entry that has a `_target_`."""
    out = []
    for _, node in (callbacks_cfg or {}).items():
        if isinstance(node, dict) and "_target_" in node:
            out.append(_instantiate(node))
    return out


def instantiate_loggers(logger_cfg) -> List[Any]:
    """instantiators.py:36-56."""
    out = []
    for _, node in (logger_cfg or {}).items():
        if isinstance(node, dict) and "_target_" in node:
            out.append(_instantiate(node))
    return out


def log_hyperparameters(object_dict: Dict[str, Any]) -> None:
    """logging_utils.py:11-57: parameter counts to the loggers (none are configured by default)."""
    model = object_dict["model"]
    n = sum(p.numel() for p in model.parameters())
    for lg in object_dict.get("logger") or []:
        if hasattr(lg, "log_hyperparams"):
            lg.log_hyperparams({"model/params/total": n})
