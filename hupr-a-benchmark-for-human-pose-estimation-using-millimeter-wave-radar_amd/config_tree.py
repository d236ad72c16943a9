"""YAML -> nested attribute object, the way the reference's main.py:7-13 builds ``cfg``."""
import os

import yaml

CONFIG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "config")


class obj(object):
    def __init__(self, d):
        for a, b in d.items():
            if isinstance(b, (list, tuple)):
                setattr(self, a, [obj(x) if isinstance(x, dict) else x for x in b])
            else:
                setattr(self, a, obj(b) if isinstance(b, dict) else b)


def load_config(name="mscsa_prgcn.yaml", config_dir=None):
    with open(os.path.join(config_dir or CONFIG_DIR, name), "r") as f:
        return obj(yaml.safe_load(f))
