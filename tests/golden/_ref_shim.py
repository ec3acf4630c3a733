"""Import the *reference* APLA modules in place from /root/reference (build container only).

This file is fixture-generation tooling: it is used by ``make_golden.py`` in the build
container to produce the ``*.npz`` / ``*.json`` vectors committed next to it.  Nothing in the
product, in ``-m gpu`` tests, in ``smoke()`` or in ``bench.py`` imports it, and the reference
sources themselves never travel with this repo.

The reference does ``from utils import *`` which drags in torchvision/easydict/timm/wandb; none
are installed.  We therefore pre-seed ``sys.modules`` with a minimal fake ``utils`` package that
exposes only what the four hot-path files touch (print_ddp, helpfuns.load_json, colours and a
stub ``download_weights``) and then load the real files by path.
"""
import importlib.util
import json
import os
import sys
import types

REF_SRC = "/root/reference/src"


def _load(name, path, package=None):
    spec = importlib.util.spec_from_file_location(name, path, submodule_search_locations=None)
    mod = importlib.util.module_from_spec(spec)
    if package is not None:
        mod.__package__ = package
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference(quiet=True):
    """Returns (vit_module, appla_attn_module, apla_vit_module, mem_eff_module)."""
    if not os.path.isdir(REF_SRC):
        raise RuntimeError("reference tree not present; goldens can only be regenerated in the build container")

    utils = types.ModuleType("utils")
    utils.__path__ = []  # mark as package
    utils.print_ddp = (lambda *_a, **_k: None) if quiet else print
    sys.modules["utils"] = utils

    colors = _load("utils.colors", os.path.join(REF_SRC, "utils/colors.py"))
    utils.colors = colors
    for k in dir(colors):
        if not k.startswith("_"):
            setattr(utils, k, getattr(colors, k))

    helpfuns = types.ModuleType("utils.helpfuns")

    def load_json(fname):
        with open(os.path.abspath(fname), "r") as f:
            return json.load(f)

    helpfuns.load_json = load_json
    sys.modules["utils.helpfuns"] = helpfuns
    utils.helpfuns = helpfuns
    utils.__all__ = ["print_ddp", "helpfuns"]

    tr = types.ModuleType("utils.transformers")
    tr.__path__ = []
    sys.modules["utils.transformers"] = tr
    tu = types.ModuleType("utils.transformers.transformers_utils")

    def download_weights(*_a, **_k):
        raise RuntimeError("no network: pretrained weights unavailable")

    tu.download_weights = download_weights
    sys.modules["utils.transformers.transformers_utils"] = tu
    vit = _load("utils.transformers.vit", os.path.join(REF_SRC, "utils/transformers/vit.py"),
                package="utils.transformers")
    tr.vit = vit

    apla_pkg = types.ModuleType("apla")
    apla_pkg.__path__ = [os.path.join(REF_SRC, "apla")]
    sys.modules["apla"] = apla_pkg
    attn = _load("apla.appla_attn", os.path.join(REF_SRC, "apla/appla_attn.py"), package="apla")
    mem = _load("apla.appla_attn_mem_eff", os.path.join(REF_SRC, "apla/appla_attn_mem_eff.py"), package="apla")
    avit = _load("apla.apla_vit", os.path.join(REF_SRC, "apla/apla_vit.py"), package="apla")
    return vit, attn, avit, mem


class Cfg(dict):
    """EasyDict-like: attribute access, AttributeError on missing keys, ``in`` works."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)
