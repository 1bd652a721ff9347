"""Model construction / checkpoint ingestion with the reference's signatures (utils.py:61-72, :326-400),
without Lightning: a `.ckpt` is a plain torch-serialised dict whose 'state_dict' holds `ss_model.*` keys."""
from __future__ import annotations

import os
from typing import Dict

import torch
import torch.nn as nn
import yaml

from .audiosep import AudioSep, get_model_class
from .metrics import calculate_sdr, calculate_sisdr  # noqa: F401  (same import surface as the reference's utils)


_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def resolve_config(config_yaml: str) -> str:
    """The reference resolves `config/audiosep_base.yaml` against the working directory (its scripts run from the repository
    root, dcase_evaluator.py:126-130).  Same here; a relative path that is not there is looked up beside the package, where
    the shipped `config/audiosep_base.yaml` lives, so `eval(evaluator, ckpt)` works from any directory."""
    if os.path.isabs(config_yaml) or os.path.exists(config_yaml):
        return config_yaml
    shipped = os.path.join(_PKG_ROOT, config_yaml)
    return shipped if os.path.exists(shipped) else config_yaml


def parse_yaml(config_yaml: str) -> Dict:
    """utils.py:61-72 (SafeLoader: the reference config holds plain scalars/lists only)."""
    with open(resolve_config(config_yaml), "r") as fr:
        return yaml.load(fr, Loader=yaml.SafeLoader)


def _build(configs: Dict) -> nn.Module:
    m = configs["model"]
    return get_model_class(model_type=m["model_type"])(input_channels=m["input_channels"],
                                                      output_channels=m["output_channels"],
                                                      condition_size=m["condition_size"])


def get_ss_model(config_yaml) -> nn.Module:
    """utils.py:326-353."""
    return _build(parse_yaml(config_yaml))


def ss_state_dict_from_checkpoint(checkpoint_path: str) -> Dict[str, torch.Tensor]:
    """Extract `ss_model.*` from a Lightning checkpoint (or accept a bare ss_model state_dict).  Loaded with
    weights_only=True: nothing in the file is executed."""
    ck = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
    sd = ck.get("state_dict", ck) if isinstance(ck, dict) else ck
    if any(k.startswith("ss_model.") for k in sd):
        sd = {k[len("ss_model."):]: v for k, v in sd.items() if k.startswith("ss_model.")}
    return sd


def load_ss_model(configs: Dict, checkpoint_path: str, query_encoder: nn.Module) -> nn.Module:
    """utils.py:356-400: returns the AudioSep holder with weights loaded (strict=False, CPU map, like the reference)."""
    ss_model = _build(configs)
    ss_model.load_state_dict(ss_state_dict_from_checkpoint(checkpoint_path), strict=False)
    return AudioSep(ss_model=ss_model, waveform_mixer=None, query_encoder=query_encoder, loss_function=None,
                    optimizer_type=None, learning_rate=None, lr_lambda_func=None)
