"""Config objects with attribute access (the reference uses EasyDict: pcdet/config.py:1-85).

`cfg_from_yaml_file` reads the same YAML layout, including `_BASE_CONFIG_` includes
(pcdet/config.py:51-58); `cfg_from_list` applies `--set KEY VALUE` overrides (:16-48)."""
import os

import yaml


class AttrDict(dict):
    """dict whose keys are attributes; nested dicts are converted on the way in."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    @staticmethod
    def _wrap(v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            return AttrDict(v)
        if isinstance(v, (list, tuple)):
            return type(v)(AttrDict._wrap(x) for x in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, AttrDict._wrap(v))

    __setattr__ = __setitem__

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


def _merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = v


def _load(path, root):
    with open(path) as f:
        raw = yaml.safe_load(f) or {}
    out = AttrDict()
    base = raw.pop("_BASE_CONFIG_", None)
    if base is not None:
        _merge(out, _load(os.path.join(root, base), root))
    for k, v in raw.items():
        if isinstance(v, dict) and "_BASE_CONFIG_" in v:
            sub = _load(os.path.join(root, v.pop("_BASE_CONFIG_")), root)
            _merge(sub, AttrDict(v))
            out[k] = sub
        elif isinstance(v, dict) and isinstance(out.get(k), dict):
            _merge(out[k], AttrDict(v))
        else:
            out[k] = v
    return out


def cfg_from_yaml_file(cfg_file, config=None, root=None):
    root = root or os.path.dirname(os.path.dirname(os.path.abspath(cfg_file)))
    loaded = _load(cfg_file, root)
    if config is None:
        return loaded
    _merge(config, loaded)
    return config


def cfg_from_list(cfg_list, config):
    """['A.B', '3', 'C', 'x'] -> config.A.B = 3 ... (values parsed with yaml)."""
    assert len(cfg_list) % 2 == 0
    for k, v in zip(cfg_list[0::2], cfg_list[1::2]):
        d = config
        parts = k.split(".")
        for p in parts[:-1]:
            d = d[p]
        assert parts[-1] in d, "NotFoundKey: %s" % k
        d[parts[-1]] = yaml.safe_load(v) if isinstance(v, str) else v
