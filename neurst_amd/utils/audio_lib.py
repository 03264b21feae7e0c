"""SpecAugment (https://arxiv.org/abs/1904.08779) as the reference applies it to pre-extracted log-mel features
(neurst/utils/audio_lib.py:24-257, numpy path `_call_numpy`): frequency masking then time masking, no time warping
(the reference only logs that it is not implemented, :93-96).

The draws use numpy's global generator in the reference's order (one `randint(0, F, size=n)` for the widths, one
`randint(0, size - f)` for the starts, per masked axis), so a run seeded like the reference masks the same cells --
tests/golden/specaug_*.npz are outputs of the reference class itself.  A mask whose start is 0 is skipped, the
feature map is modified IN PLACE and also returned (both as in the reference, :139-147).
"""
import math

import numpy
import yaml


class SpecAugment(object):
    _PREDEF_SETTINGS = {
        "LB": {"time_wrap_w": 80, "freq_mask_n": 1, "freq_mask_f": 27, "time_mask_n": 1, "time_mask_t": 100, "time_mask_p": 1.},
        "LD": {"time_wrap_w": 80, "freq_mask_n": 2, "freq_mask_f": 27, "time_mask_n": 2, "time_mask_t": 100, "time_mask_p": 1.},
        "SM": {"time_wrap_w": 40, "freq_mask_n": 2, "freq_mask_f": 15, "time_mask_n": 2, "time_mask_t": 70, "time_mask_p": 0.2},
        "SS": {"time_wrap_w": 40, "freq_mask_n": 2, "freq_mask_f": 27, "time_mask_n": 2, "time_mask_t": 70, "time_mask_p": 0.2},
    }

    def __init__(self, time_wrap_w, freq_mask_n, freq_mask_f, time_mask_n, time_mask_t, time_mask_p, mask_value=None,
                 rng=None):
        self._time_wrap_w = time_wrap_w
        self._freq_mask_n = freq_mask_n
        self._freq_mask_f = freq_mask_f
        self._time_mask_n = time_mask_n
        self._time_mask_t = time_mask_t
        self._time_mask_p = time_mask_p
        self._mask_value = mask_value
        self._rng = rng if rng is not None else numpy.random  # the reference draws from the global generator
        assert self._time_mask_t > 0
        assert self._freq_mask_f > 0

    @classmethod
    def build(cls, setting):
        """setting: None | predefined name (LB, LD, SM, SS) | yaml / dict of constructor arguments (:98-108)."""
        if setting is None:
            return None
        if isinstance(setting, str):
            setting = yaml.load(setting, Loader=yaml.FullLoader)
        if isinstance(setting, str):
            setting = cls._PREDEF_SETTINGS.get(setting, None)
        if setting is None:
            return None
        assert isinstance(setting, dict), f"Unknown type of setting: {setting}"
        return cls(**setting)

    def _mask(self, spectrogram, n, F, mask_value, axis, p=None):
        size = spectrogram.shape[axis]
        if size < F:
            return spectrogram
        if p:
            F = min(F, math.floor(size * p))
        f = self._rng.randint(0, F, size=n)
        f0 = self._rng.randint(0, size - f)
        for i in range(n):
            if f0[i] == 0:
                continue
            if axis == 0:
                spectrogram[f0[i]: f0[i] + f[i], :] = mask_value
            else:
                spectrogram[:, f0[i]: f0[i] + f[i]] = mask_value
        return spectrogram

    def __call__(self, spectrogram):
        """spectrogram: numpy array [frames, feature_dim * channels]."""
        if spectrogram.ndim > 2:
            raise ValueError("batch specaug is not implemented for numpy array.")
        mask_value = self._mask_value
        if mask_value is None:
            mask_value = spectrogram.mean()
        if self._freq_mask_n > 0:
            spectrogram = self._mask(spectrogram, self._freq_mask_n, self._freq_mask_f, mask_value, axis=1)
        if self._time_mask_n > 0:
            spectrogram = self._mask(spectrogram, self._time_mask_n, self._time_mask_t, mask_value, axis=0,
                                     p=self._time_mask_p)
        return spectrogram
