"""Configuration objects for the MI355X extraction path.

Field-for-field compatible with the reference's configuration layer
(reference: src/amcpy/config.py:15-57 ``Paths``, :60-110 ``SignalConfig``,
:113-148 ``FeatureConfig``, :179-186 ``Config``) so that a caller that builds a
``Config()`` for ``run_extraction`` can hand the very same attribute paths to
this engine:

    cfg.paths.mat_data / cfg.paths.mat_filename / cfg.paths.calculated_features
    cfg.paths.ensure_dirs()
    cfg.signals.frame_size / num_frames / num_threads / snr_values / mat_info
    cfg.signals.modulations_with_noise
    cfg.features.all_features / used

Only the extraction-relevant groups are restated.  ``TrainingConfig`` is kept
as a plain value holder (the classifier is out of scope)
so that ``Config().training`` still resolves for downstream code.

Objects are immutable (frozen dataclasses), as in the reference; custom
configurations are made with constructor keywords, e.g.
``SignalConfig(snr_values={0: "0", 1: "10"}, num_frames=500)``.
"""

from __future__ import annotations

import os
from dataclasses import dataclass, field, make_dataclass
from pathlib import Path
from typing import ClassVar, Dict, Tuple

# sub-directory name for each Paths attribute (reference config.py:36-43)
_SUBDIRS: Dict[str, str] = {
    "mat_data": "mat-data",
    "calculated_features": "calculated-features",
    "arm_data": "arm-data",
    "trained_ann": "ann",
    "figures": "figures",
    "feature_figures": "figures/features",
}

_MODS: Tuple[str, ...] = ("BPSK", "QPSK", "8PSK", "16QAM", "64QAM")
_NOISE = "WGN"

# .mat variable holding each modulation's (n_snr, n_frames, >=frame_size) array
# (reference config.py:101-110, README.md:60-73)
_MAT_VARS: Dict[str, str] = {
    "BPSK": "signal_bpsk",
    "QPSK": "signal_qpsk",
    "8PSK": "signal_8psk",
    "16QAM": "signal_qam16",
    "64QAM": "signal_qam64",
    "WGN": "signal_noise",
}


def _default_snr_grid() -> Dict[int, str]:
    # 16 entries, -10 dB .. +20 dB in 2 dB steps (reference config.py:75-94)
    return {i: str(-10 + 2 * i) for i in range(16)}


@dataclass(frozen=True)
class Paths:
    """Directory layout; every directory hangs off ``root``."""

    root: Path = field(default_factory=lambda: Path(os.getcwd()))
    mat_data: Path = field(init=False)
    calculated_features: Path = field(init=False)
    arm_data: Path = field(init=False)
    trained_ann: Path = field(init=False)
    figures: Path = field(init=False)
    feature_figures: Path = field(init=False)
    mat_filename: str = "all_modulations.mat"

    def __post_init__(self) -> None:
        base = Path(self.root)
        object.__setattr__(self, "root", base)
        for attr, sub in _SUBDIRS.items():
            object.__setattr__(self, attr, base / sub)

    def ensure_dirs(self) -> None:
        """Create every directory of the layout that does not exist yet."""
        for attr in _SUBDIRS:
            getattr(self, attr).mkdir(parents=True, exist_ok=True)


@dataclass(frozen=True)
class SignalConfig:
    """What the IQ container holds: modulations, SNR grid, frame geometry."""

    modulations: Tuple[str, ...] = _MODS
    modulations_with_noise: Tuple[str, ...] = _MODS + (_NOISE,)
    labels: Tuple[int, ...] = tuple(range(len(_MODS) + 1))
    snr_values: Dict[int, str] = field(default_factory=_default_snr_grid)
    frame_size: int = 2048
    num_frames: int = 1000
    # Advisory here: the reference starts this many Python threads per
    # modulation process (feature_extraction.py:58-61); the HIP path has no
    # host worker threads, the GPU walks all frames of a shard in one launch.
    num_threads: int = 8
    mat_info: Dict[str, str] = field(default_factory=lambda: dict(_MAT_VARS))


@dataclass(frozen=True)
class FeatureConfig:
    """The 18 feature ids, their display names and the classifier's subset."""

    names: ClassVar[Dict[int, str]] = {
        i + 1: f"${body}$" for i, body in enumerate(
            [r"\gamma_{max}", r"\sigma_{ap}", r"\sigma_{dp}", r"\sigma_{aa}", r"\sigma_{af}", "X", "X_2",
             r"\mu_{42}^{a}", r"\mu_{42}^{f}"]
            + [f"C_{{{pq}}}" for pq in ("20", "21", "40", "41", "42", "60", "61", "62", "63")])
    }

    all_features: Tuple[int, ...] = tuple(range(1, 19))
    used: Tuple[int, ...] = (2, 4, 6, 8, 12, 14)

    @property
    def used_names(self):
        return [self.names[i] for i in self.used]

    @property
    def num_used(self) -> int:
        return len(self.used)


def _feature_files(self):
    return [f"{m}_features" for m in SignalConfig().modulations_with_noise]


# Classifier hyper-parameters: carried for attribute compatibility only (reference
# config.py:151-176); nothing on the extraction path reads them, so the group is
# declared as data rather than spelled out as a class body.
_TRAINING_DEFAULTS = (
    ("training_snr", Tuple[int, ...], tuple(range(10, 16))),
    ("all_snr", Tuple[int, ...], tuple(range(16))),
    ("plotting_snr", Tuple[int, ...], tuple(range(16))),
    ("test_size", float, 0.2), ("random_state", int, 42),
    ("activation", str, "relu"), ("batch_size", int, 128), ("dropout", float, 0.4),
    ("epochs", int, 21), ("learning_rate", float, 0.001418378071933655),
    ("optimizer", str, "rmsprop"),
    ("layer_size_hl1", int, 26), ("layer_size_hl2", int, 29), ("layer_size_hl3", int, 30),
)
TrainingConfig = make_dataclass(
    "TrainingConfig", [(n, t, field(default=d)) for n, t, d in _TRAINING_DEFAULTS], frozen=True,
    namespace={"feature_files": property(_feature_files),
               "__doc__": "Classifier hyper-parameters (value holder; out of the extraction scope)."})


@dataclass(frozen=True)
class Config:
    """Top-level bundle, same four groups as the reference (config.py:179-186)."""

    paths: Paths = field(default_factory=Paths)
    signals: SignalConfig = field(default_factory=SignalConfig)
    features: FeatureConfig = field(default_factory=FeatureConfig)
    training: TrainingConfig = field(default_factory=TrainingConfig)
