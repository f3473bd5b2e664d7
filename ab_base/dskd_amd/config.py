"""mmcv-free loader for the reference's python config files (ext-mmcv ``Config.fromfile``
as used at /root/reference/tools/train_increment.py:107-113): executes the file, merges the
``_base_`` chain (child keys win, dicts merge recursively, ``_delete_=True`` replaces), and
gives attribute access.  ``merge_from_dict`` implements ``--cfg-options a.b=c``."""
import ast
import copy
import os


class ConfigDict(dict):
    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value

    def __deepcopy__(self, memo):
        return ConfigDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


def _wrap(obj):
    if isinstance(obj, dict):
        return ConfigDict({k: _wrap(v) for k, v in obj.items()})
    if isinstance(obj, list):
        return [_wrap(v) for v in obj]
    if isinstance(obj, tuple):
        return tuple(_wrap(v) for v in obj)
    return obj


def _merge(child, base):
    out = dict(base)
    for k, v in child.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict) and not v.get("_delete_", False):
            out[k] = _merge(v, out[k])
        else:
            if isinstance(v, dict):
                v = {kk: vv for kk, vv in v.items() if kk != "_delete_"}
            out[k] = v
    return out


def _exec_file(path):
    with open(path, "r", encoding="utf-8") as f:
        src = f.read()
    ast.parse(src)
    scope = {"__file__": path, "__name__": "__dskd_config__"}
    exec(compile(src, path, "exec"), scope)
    import types
    return {k: v for k, v in scope.items()
            if not k.startswith("__") and not isinstance(v, (types.ModuleType, types.FunctionType))}


def _load(path):
    path = os.path.abspath(os.path.expanduser(path))
    if not os.path.isfile(path):
        raise FileNotFoundError(path)
    cfg = _exec_file(path)
    bases = cfg.pop("_base_", [])
    if isinstance(bases, str):
        bases = [bases]
    merged = {}
    for b in bases:
        bcfg = _load(os.path.join(os.path.dirname(path), b))
        dup = set(merged) & set(bcfg)
        if dup:
            raise KeyError(f"Duplicate key is not allowed among bases: {sorted(dup)}")
        merged.update(bcfg)
    return _merge(cfg, merged)


class Config:
    def __init__(self, cfg_dict=None, filename=None):
        object.__setattr__(self, "_cfg_dict", _wrap(cfg_dict or {}))
        object.__setattr__(self, "filename", filename)

    @staticmethod
    def fromfile(filename):
        return Config(_load(filename), filename=os.path.abspath(filename))

    def __getattr__(self, name):
        return getattr(self._cfg_dict, name)

    def __setattr__(self, name, value):
        self._cfg_dict[name] = _wrap(value)

    def __getitem__(self, name):
        return self._cfg_dict[name]

    def __setitem__(self, name, value):
        self._cfg_dict[name] = _wrap(value)

    def __contains__(self, name):
        return name in self._cfg_dict

    def get(self, key, default=None):
        return self._cfg_dict.get(key, default)

    def keys(self):
        return self._cfg_dict.keys()

    def to_dict(self):
        return copy.deepcopy(dict(self._cfg_dict))

    def merge_from_dict(self, options):
        """options: {'a.b.c': value}; list indices allowed (``data.train.0.x``)."""
        for full_key, v in options.items():
            d = self._cfg_dict
            keys = full_key.split(".")
            for k in keys[:-1]:
                if isinstance(d, list):
                    d = d[int(k)]
                else:
                    d = d.setdefault(k, ConfigDict())
            last = keys[-1]
            if isinstance(d, list):
                d[int(last)] = _wrap(v)
            else:
                d[last] = _wrap(v)
