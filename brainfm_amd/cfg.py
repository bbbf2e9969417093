"""YAML -> nested Namespace, the way the reference's utils/config.py:70-103 +
utils/process_cfg.py:54-68 + utils/misc.py:378-381,611-617 do it: files are
merged left to right, nested mappings recursively, lists replaced.  The
reference's out_dir stamping (process_cfg.py:9-29) is deliberately left out --
it only matters to the training launcher."""
import os
import re
from argparse import Namespace

import yaml

_FLOAT = re.compile(u'''^(?:
    [-+]?(?:[0-9][0-9_]*)\\.[0-9_]*(?:[eE][-+]?[0-9]+)?
    |[-+]?(?:[0-9][0-9_]*)(?:[eE][-+]?[0-9]+)
    |\\.[0-9_]+(?:[eE][-+][0-9]+)?
    |[-+]?[0-9][0-9_]*(?::[0-5]?[0-9])+\\.[0-9_]*
    |[-+]?\\.(?:inf|Inf|INF)
    |\\.(?:nan|NaN|NAN))$''', re.X)


class _Loader(yaml.SafeLoader):
    pass


_Loader.add_implicit_resolver(u'tag:yaml.org,2002:float', _FLOAT, list(u'-+0123456789.'))


def _merge(dst, src):
    if src is None:
        return dst
    for k, v in src.items():
        if isinstance(v, dict):
            dst[k] = _merge(dst[k] if isinstance(dst.get(k), dict) else {}, v)
        else:
            dst[k] = v
    return dst


def load_config(default_cfg_file, add_cfg_files=(), cfg_dir=""):
    cfg = {}
    files = [default_cfg_file] + [f for f in add_cfg_files if f]
    for i, f in enumerate(files):
        if not os.path.isabs(f) and i > 0 and cfg_dir:
            f = os.path.join(cfg_dir, f if f.endswith(".yaml") else f + ".yaml")
        if not os.path.exists(f):
            raise ValueError("Provided config path not existed: %s" % f)
        with open(f, "r") as fh:
            _merge(cfg, yaml.load(fh, Loader=_Loader))
    return cfg


def nested_dict_to_namespace(d):
    if isinstance(d, dict):
        return Namespace(**{k: nested_dict_to_namespace(v) for k, v in d.items()})
    return d


def preprocess_cfg(cfg_files, cfg_dir=""):
    cfg_files = [f for f in cfg_files if f]
    return nested_dict_to_namespace(load_config(cfg_files[0], cfg_files[1:], cfg_dir))
