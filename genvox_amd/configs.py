"""Configuration objects for the MI355X Tacotron2 forward path.

Host-side mirror of the three config sections the reference reads when it builds a
model for inference (reference: configs/__init__.py:16-59 BaseConfig file I/O,
:61-82 TextConfig, :92-167 AudioConfig, configs/models.py:4-86 Tacotron2Config).
Only the attribute names, defaults, accepted ranges and the on-disk layout
(one mapping per section: ``model_config``, ``audio_config``, ``text_config``)
are kept; the implementation is table driven and has no dependency on the
reference's ``utils`` package (the reference's configs import a YouTube
downloader through it).
"""
from __future__ import annotations

import json
import os
from typing import Any, Dict, Mapping, Optional, Tuple

import yaml

_Range = Tuple[Optional[float], Optional[float]]


def check_argument(name: str, value, min_val=None, max_val=None) -> None:
    """Range assert with the reference's error behaviour (configs/__init__.py:7-13):
    an AssertionError naming the offending field."""
    if min_val is not None and max_val is not None:
        assert min_val <= value <= max_val, (
            f"The value '{name}' ({value}) is not in the required range ({min_val} -> {max_val}).")
    elif min_val is not None:
        assert value >= min_val, f"The value '{name}' ({value}) is below min_val ({min_val})."
    elif max_val is not None:
        assert value <= max_val, f"The value '{name}' ({value}) is above max_val ({max_val})."


class BaseConfig:
    """Field-table driven config. Subclasses declare ``_FIELDS = {name: (default, (lo, hi))}``."""

    _FIELDS: Dict[str, Tuple[Any, Optional[_Range]]] = {}

    def __init__(self, **kwargs):
        unknown = set(kwargs) - set(self._FIELDS)
        if unknown:
            raise TypeError(f"{type(self).__name__}() got unexpected argument(s): {sorted(unknown)}")
        for name, (default, _rng) in self._FIELDS.items():
            setattr(self, name, kwargs.get(name, default))
        self._normalise()
        for name, (_default, rng) in self._FIELDS.items():
            if rng is not None:
                lo, hi = self._resolve_range(name, rng)
                check_argument(name, getattr(self, name), min_val=lo, max_val=hi)

    def _normalise(self) -> None:
        pass

    def _resolve_range(self, name: str, rng: _Range) -> _Range:
        return rng

    # ---- presentation -------------------------------------------------
    def __str__(self) -> str:
        items = list(vars(self).items())
        lines = [type(self).__name__]
        for i, (k, v) in enumerate(items):
            lines.append(("└── " if i == len(items) - 1 else "├── ") + k.ljust(35) + f"({v})")
        return "\n".join(lines)

    def __repr__(self) -> str:
        return f"{type(self).__name__}()"

    def to_dict(self) -> Dict[str, Any]:
        return {k: v for k, v in vars(self).items() if not isinstance(v, BaseConfig)}

    # ---- file I/O (same section layout as the reference's exp/config.yaml) ----
    @staticmethod
    def write_configs_to_file(path: str, configs: Mapping[str, Optional["BaseConfig"]]) -> None:
        ext = os.path.splitext(path)[1][1:]
        assert ext in ("json", "yaml"), f"given config extension ({ext}) is invalid"
        blob = {name: cfg.to_dict() for name, cfg in configs.items() if cfg is not None}
        with open(path, "w") as f:
            if ext == "json":
                json.dump(blob, f, indent=4)
            else:
                yaml.dump(blob, f, sort_keys=False, allow_unicode=True)

    @staticmethod
    def load_configs_from_file(path: str, config_map: Mapping[str, type]) -> Dict[str, "BaseConfig"]:
        ext = os.path.splitext(path)[1][1:]
        assert ext in ("json", "yaml"), f"given config extension ({ext}) is invalid"
        with open(path, "r") as f:
            blob = json.load(f) if ext == "json" else yaml.load(f, Loader=yaml.SafeLoader)
        return {name: config_map[name](**section) for name, section in blob.items() if name in config_map}


class TextConfig(BaseConfig):
    """reference: configs/__init__.py:61-82. ``n_tokens`` sizes the embedding table."""

    _FIELDS = {
        "language": ("english", None),
        "cleaners": (None, None),
        "use_g2p": (False, None),
        "token_map": (None, None),
        "n_tokens": (None, None),
    }

    def _normalise(self) -> None:
        self.language = self.language.lower()


class AudioConfig(BaseConfig):
    """reference: configs/__init__.py:92-167 (same names, defaults and ranges)."""

    _FIELDS = {
        "sampling_rate": (22050, (16000, 44100)),
        "trim_silence": (True, None),
        "trim_dbfs": (-50.0, (-100, 0)),
        "min_wav_duration": (0.5, (0.1, None)),
        "max_wav_duration": (10, ("min_wav_duration", None)),
        "normalize": (True, None),
        "filter_length": (512, (256, 2048)),
        "hop_length": (256, (128, "filter_length")),
        "n_mels": (80, (12, 128)),
        "mel_fmin": (0.0, (0, 8000)),
        "mel_fmax": (8000.0, (8000, 22050)),
        "log_func": ("np.log10", None),
        "ref_level_db": (1.0, (1, None)),
    }

    def _resolve_range(self, name, rng):
        return tuple(getattr(self, b) if isinstance(b, str) else b for b in rng)


class Tacotron2Config(BaseConfig):
    """reference: configs/models.py:4-86 (same names, defaults and ranges)."""

    _FIELDS = {
        "symbols_embedding_dim": (512, (1, None)),
        "encoder_kernel_size": (5, (1, None)),
        "encoder_n_convolutions": (3, (1, None)),
        "encoder_embedding_dim": (512, (1, None)),
        "decoder_rnn_dim": (1024, (1, None)),
        "prenet_dim": (256, (1, None)),
        "max_decoder_steps": (1000, (1, 10000)),
        "gate_threshold": (0.5, (0, 1)),
        "p_attention_dropout": (0.1, (0, None)),
        "p_decoder_dropout": (0.1, (0, None)),
        "attention_rnn_dim": (1024, (1, None)),
        "attention_dim": (128, (1, None)),
        "attention_location_n_filters": (32, (1, None)),
        "attention_location_kernel_size": (31, (1, None)),
        "postnet_embedding_dim": (512, (1, None)),
        "postnet_kernel_size": (5, (1, None)),
        "postnet_n_convolutions": (5, (1, None)),
        "mask_padding": (True, None),
        "learning_rate": (1e-3, (1e-5, None)),
        "weight_decay": (1e-6, (0, None)),
        "grad_clip_thresh": (1.0, (0, None)),
        "beta1": (0.9, (0, 1)),
        "beta2": (0.999, (0, 1)),
    }
