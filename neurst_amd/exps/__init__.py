from neurst_amd.exps.base_experiment import BaseExperiment, build_exp, register_exp  # noqa: F401
from neurst_amd.exps import trainer  # noqa: F401
from neurst_amd.exps import sequence_generator  # noqa: F401
from neurst_amd.exps import evaluator  # noqa: F401
