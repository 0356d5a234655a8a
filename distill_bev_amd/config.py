"""Minimal ``mmcv.Config`` replacement: python-dict config files with ``_base_`` inheritance,
``_delete_`` and ``--cfg-options``-style dotted overrides (what ``tools/train.py:63-72,105-107``
and ``scripts/teacher_to_bevdepth4d/centerpoint2bevdepth.sh:23-47`` rely on), so that the
reference's own config files load unchanged without mmcv.
"""
import ast
import copy
import os

BASE_KEY = "_base_"
DELETE_KEY = "_delete_"


class ConfigDict(dict):
    """dict with attribute access (addict-like, as mmcv's ConfigDict)."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError as e:
            raise AttributeError(name) from e

    def __setattr__(self, name, value):
        self[name] = value

    def __deepcopy__(self, memo):
        return ConfigDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


def _to_cfgdict(x):
    if isinstance(x, dict):
        return ConfigDict({k: _to_cfgdict(v) for k, v in x.items()})
    if isinstance(x, list):
        return [_to_cfgdict(v) for v in x]
    if isinstance(x, tuple):
        return tuple(_to_cfgdict(v) for v in x)
    return x


def _merge_a_into_b(a, b):
    """mmcv Config._merge_a_into_b: dicts merge recursively, ``_delete_=True`` replaces."""
    b = copy.deepcopy(b)
    for k, v in a.items():
        if isinstance(v, dict) and k in b and isinstance(b[k], dict) and not v.get(DELETE_KEY, False):
            b[k] = _merge_a_into_b(v, b[k])
        elif isinstance(v, dict):
            v = {kk: vv for kk, vv in v.items() if kk != DELETE_KEY}
            b[k] = copy.deepcopy(v)
        else:
            b[k] = copy.deepcopy(v)
    return b


def _exec_file(path):
    with open(path, "r") as f:
        src = f.read()
    scope = {"__file__": path}
    exec(compile(src, path, "exec"), scope)  # config files are python (mmcv does the same)
    return {k: v for k, v in scope.items()
            if not k.startswith("__") and not callable(v) and not isinstance(v, type(os))}


def _load(path):
    path = os.path.abspath(path)
    cfg = _exec_file(path)
    bases = cfg.pop(BASE_KEY, None)
    if bases is None:
        return cfg
    if isinstance(bases, str):
        bases = [bases]
    merged = {}
    for b in bases:
        bcfg = _load(os.path.join(os.path.dirname(path), b))
        dup = set(merged) & set(bcfg)
        if dup:
            raise KeyError(f"duplicate keys in bases of {path}: {sorted(dup)}")
        merged.update(bcfg)
    return _merge_a_into_b(cfg, merged)


def _parse_value(v):
    """--cfg-options value syntax (mmcv DictAction): ints, floats, bools, None, [a,b], (a,b), 'str'."""
    if not isinstance(v, str):
        return v
    s = v.strip()
    try:
        return ast.literal_eval(s)
    except (ValueError, SyntaxError):
        pass
    if s.lower() in ("true", "false"):
        return s.lower() == "true"
    if s.lower() in ("none", "null"):
        return None
    if (s.startswith("[") and s.endswith("]")) or (s.startswith("(") and s.endswith(")")):
        inner = s[1:-1]
        items = [_parse_value(p) for p in inner.split(",") if p.strip() != ""]
        return items if s.startswith("[") else tuple(items)
    return s


class Config:
    def __init__(self, cfg_dict=None, filename=None):
        object.__setattr__(self, "_cfg", _to_cfgdict(cfg_dict or {}))
        object.__setattr__(self, "filename", filename)

    @staticmethod
    def fromfile(filename):
        return Config(_load(filename), filename)

    def merge_from_dict(self, options):
        """dotted-key overrides, e.g. {'model.distill_params.fp_weight': 6e-2}."""
        nested = {}
        for full_key, v in options.items():
            d = nested
            keys = full_key.split(".")
            for k in keys[:-1]:
                d = d.setdefault(k, {})
            d[keys[-1]] = _parse_value(v)
        object.__setattr__(self, "_cfg", _to_cfgdict(_merge_a_into_b(nested, self._cfg)))

    def merge_from_args(self, args):
        """list of 'a.b.c=value' strings as after --cfg-options."""
        opts = {}
        for a in args:
            k, v = a.split("=", 1)
            opts[k] = v
        self.merge_from_dict(opts)

    def __getattr__(self, name):
        return getattr(self._cfg, name)

    def __getitem__(self, name):
        return self._cfg[name]

    def __contains__(self, name):
        return name in self._cfg

    def get(self, k, default=None):
        return self._cfg.get(k, default)

    def to_dict(self):
        return copy.deepcopy(dict(self._cfg))
