"""Metric base class (neurst/metrics/metric.py:20-75): a callable hypothesis list (+ references) -> result dict, a `flag`
naming the headline entry, and the comparison used to keep the best checkpoints."""
import numpy as np


class Metric(object):
    REGISTRY_NAME = "metric"

    def __init__(self, *args, **kwargs):
        self._flag = self.__class__.__name__

    @property
    def flag(self):
        return self._flag

    @flag.setter
    def flag(self, flag_name):
        self._flag = flag_name

    def set_groundtruth(self, groundtruth):
        pass

    def greater_or_eq(self, result1, result2):
        return self.get_value(result1) >= self.get_value(result2)

    def get_value(self, result):
        if isinstance(result, (float, np.floating)):
            return float(result)
        return result[self._flag]

    def __call__(self, hypothesis, groundtruth=None):
        return self.call(hypothesis, groundtruth)

    def call(self, hypothesis, groundtruth=None):
        raise NotImplementedError
