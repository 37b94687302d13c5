"""The subset of Hydra / OmegaConf that `src/train.py experiment=pretraining_medmoe` needs, for images without `hydra-core`
(this one has PyYAML only).  With hydra installed, src/train.py uses the real thing and this module is not imported.

Supported (what the reference's config tree uses, configs/train.yaml:5-27, configs/experiment/pretraining_medmoe.yaml:6-11,
configs/model/med-moe_pretraining.yaml:1-3):
  * defaults lists: `group: name`, `/group: name`, `name` / `name.yaml` (same directory), `_self_`, `optional group: name`,
    `override /group: name`, `group: null`
  * `# @package _global_` headers (experiment files merge at the root; group files merge under their group path)
  * command-line overrides `a.b.c=value`, `group=name`, `+a.b=value`
  * interpolation `${a.b}`, `${oc.env:VAR}`, `${oc.env:VAR,default}`, `${hydra:runtime.output_dir}`, `${hydra:runtime.cwd}`, `${now:FMT}`
  * `instantiate(cfg, **kwargs)`: `_target_` by importlib, `_partial_: true` -> functools.partial, recursive into nested dicts,
    `_recursive_: false`
Dict nodes are `Cfg` (a dict with attribute access and `.get`), so `cfg.model._target_` and `loss_cfg.get("temp3")` read as
they do on a DictConfig.
"""
import datetime
import functools
import importlib
import os
import re
from typing import Any, Dict, List, Optional

import yaml


class Cfg(dict):
    """dict with attribute access (read and write), the DictConfig surface the mirror modules touch."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k) from None

    def __setattr__(self, k, v):
        self[k] = v


_FLOAT = re.compile(r"^[+-]?(\d+\.?\d*|\.\d+)[eE][+-]?\d+$")


def _wrap(x):
    if isinstance(x, str) and _FLOAT.match(x):
        return float(x)             # PyYAML (YAML 1.1) reads `1e-6` as a string; OmegaConf / Hydra read a float
    if isinstance(x, dict):
        return Cfg({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, list):
        return [_wrap(v) for v in x]
    return x


def _merge(dst: dict, src: dict) -> dict:
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = _wrap(v)
    return dst


def _set_path(root: dict, dotted: str, value, create=True):
    keys = [k for k in dotted.split(".") if k]
    node = root
    for k in keys[:-1]:
        if k not in node or not isinstance(node[k], dict):
            if not create:
                raise KeyError(dotted)
            node[k] = Cfg()
        node = node[k]
    node[keys[-1]] = _wrap(value)


def _get_path(root: dict, dotted: str):
    node = root
    for k in dotted.split("."):
        node = node[int(k)] if isinstance(node, list) else node[k]
    return node


class _Composer:
    def __init__(self, config_dir: str, group_choices: Dict[str, Optional[str]]):
        self.dir = config_dir
        self.choices = dict(group_choices)        # command-line `group=name` and `override /group: name` selections
        self.loaded: List[str] = []

    def _read(self, rel: str):
        path = os.path.join(self.dir, rel if rel.endswith(".yaml") else rel + ".yaml")
        with open(path) as f:
            text = f.read()
        head = text.lstrip().splitlines()[0] if text.strip() else ""
        m = re.match(r"#\s*@package\s+(\S+)", head)
        self.loaded.append(os.path.relpath(path, self.dir))
        return (yaml.safe_load(text) or {}), (m.group(1) if m else None)

    def compose_file(self, rel: str, package: str, out: dict):
        """Merge config file `rel` (relative to the config dir) into `out` under `package` ('' = root)."""
        body, pkg_decl = self._read(rel)
        if pkg_decl == "_global_":
            package = ""
        defaults = body.pop("defaults", None) or ["_self_"]
        if "_self_" not in defaults:
            defaults = list(defaults) + ["_self_"]           # hydra >= 1.1 appends _self_ when it is not listed
        here = os.path.dirname(rel)
        # first pass: `override` entries only record a choice
        for d in defaults:
            if isinstance(d, dict):
                (k, v), = d.items()
                if k.startswith("override "):
                    g = k[len("override "):].strip().lstrip("/")
                    self.choices.setdefault(g, v)
        for d in defaults:
            if d == "_self_":
                target = out
                if package:
                    node = out
                    for k in package.split("."):
                        node = node.setdefault(k, Cfg())
                    target = node
                _merge(target, body)
                continue
            if isinstance(d, str):                          # sibling file
                self.compose_file(os.path.join(here, d), package, out)
                continue
            (k, v), = d.items()
            if k.startswith("override "):
                continue
            optional = k.startswith("optional ")
            if optional:
                k = k[len("optional "):].strip()
            absolute = k.startswith("/")
            g = k.lstrip("/")
            gdir = g if absolute or not here else os.path.join(here, g)
            v = self.choices.get(g, v)
            if v is None or v == "null":
                continue
            rel2 = os.path.join(gdir, str(v))
            if not os.path.exists(os.path.join(self.dir, rel2 if rel2.endswith(".yaml") else rel2 + ".yaml")):
                if optional or g.startswith("hydra/"):
                    continue                                 # hydra's own launcher / logging groups live inside hydra-core
                raise FileNotFoundError(f"config group file {rel2}.yaml not found under {self.dir}")
            self.compose_file(rel2, g.replace("/", "."), out)


_INTERP = re.compile(r"\$\{([^${}]+)\}")


def _resolve_value(s: str, root: dict, runtime: Dict[str, str]):
    def one(expr: str):
        expr = expr.strip()
        if expr.startswith("oc.env:"):
            name, _, default = expr[len("oc.env:"):].partition(",")
            if name in os.environ:
                return os.environ[name]
            if default != "":
                return default
            raise KeyError(f"environment variable {name} is not set (needed by the config)")
        if expr.startswith("hydra:"):
            return runtime[expr[len("hydra:"):]]
        if expr.startswith("now:"):
            return datetime.datetime.now().strftime(expr[len("now:"):])
        v = _get_path(root, expr)
        return _resolve_value(v, root, runtime) if isinstance(v, str) else v

    m = _INTERP.fullmatch(s.strip())
    if m:                                                    # whole-string interpolation keeps the referenced type
        return one(m.group(1))
    prev = None
    while prev != s and _INTERP.search(s):
        prev = s
        s = _INTERP.sub(lambda mm: str(one(mm.group(1))), s)
    return s


def _resolve(node, root, runtime):
    if isinstance(node, dict):
        for k in list(node.keys()):
            node[k] = _resolve(node[k], root, runtime)
        return node
    if isinstance(node, list):
        return [_resolve(v, root, runtime) for v in node]
    if isinstance(node, str) and "${" in node:
        return _resolve_value(node, root, runtime)
    return node


def compose(config_dir: str, config_name: str = "train.yaml", overrides: Optional[List[str]] = None, resolve: bool = True,
            output_dir: Optional[str] = None) -> Cfg:
    """Compose `config_name` with command-line style `overrides` (hydra.main semantics for the supported subset)."""
    overrides = list(overrides or [])
    groups = {d for d in os.listdir(config_dir) if os.path.isdir(os.path.join(config_dir, d))}
    choices, values = {}, []
    for ov in overrides:
        key, eq, val = ov.partition("=")
        if not eq:
            raise ValueError(f"override '{ov}' is not key=value")
        key = key.lstrip("+")
        if key in groups or key.lstrip("/") in groups:
            choices[key.lstrip("/")] = None if val in ("null", "") else val
        else:
            values.append((key, yaml.safe_load(val)))
    comp = _Composer(config_dir, choices)
    out = Cfg()
    comp.compose_file(config_name, "", out)
    for key, val in values:
        _set_path(out, key, val)
    if resolve:
        cwd = os.getcwd()
        runtime = {"runtime.cwd": cwd, "runtime.output_dir": output_dir or os.path.join(cwd, "outputs")}
        _resolve(out, out, runtime)
    out["_composed_from_"] = comp.loaded
    return out


# ---------------------------------------------------------------------------------------------------------------------
# instantiate
# ---------------------------------------------------------------------------------------------------------------------
# A `_target_` whose package is absent here resolves to the build's stand-in with the same constructor surface.
TARGET_FALLBACKS = {
    "lightning.pytorch.trainer.Trainer": "medmoe_amd.trainer.Trainer",
    "lightning.pytorch.callbacks.ModelCheckpoint": "medmoe_amd.trainer.ModelCheckpoint",
    "lightning.pytorch.callbacks.EarlyStopping": "medmoe_amd.trainer.EarlyStopping",
}


def locate(target: str):
    """Dotted path -> object (hydra.utils.get_class / get_method)."""
    def load(path):
        mod, _, attr = path.rpartition(".")
        obj = importlib.import_module(mod)
        return getattr(obj, attr)
    try:
        return load(target)
    except (ImportError, AttributeError):
        if target in TARGET_FALLBACKS:
            return load(TARGET_FALLBACKS[target])
        raise


def targets_of(cfg) -> List[str]:
    """Every `_target_` string in a config tree (depth first)."""
    out = []
    if isinstance(cfg, dict):
        if "_target_" in cfg:
            out.append(cfg["_target_"])
        for v in cfg.values():
            out += targets_of(v)
    elif isinstance(cfg, list):
        for v in cfg:
            out += targets_of(v)
    return out


def instantiate(cfg, *args, **kwargs):
    """hydra.utils.instantiate for the supported subset: nested `_target_` nodes are built first (unless
    `_recursive_: false`), `_partial_: true` returns a functools.partial, keyword arguments override config keys."""
    if not isinstance(cfg, dict) or "_target_" not in cfg:
        return cfg
    recursive = cfg.get("_recursive_", True)
    partial = bool(cfg.get("_partial_", False))

    def build(v):
        if isinstance(v, dict):
            if "_target_" in v and recursive:
                return instantiate(v)
            return Cfg({k: build(x) for k, x in v.items()})
        if isinstance(v, list):
            return [build(x) for x in v]
        return v

    params = {k: build(v) for k, v in cfg.items() if k not in ("_target_", "_partial_", "_recursive_", "_convert_")}
    params.update(kwargs)
    fn = locate(cfg["_target_"])
    return functools.partial(fn, *args, **params) if partial else fn(*args, **params)
