from neurst_amd.data.datasets.dataset import Dataset, build_dataset, register_dataset  # noqa: F401
from neurst_amd.data.datasets import synthetic_speech  # noqa: F401
from neurst_amd.data.datasets import synthetic_text  # noqa: F401
from neurst_amd.data.datasets import audio_dataset  # noqa: F401
from neurst_amd.data.datasets import parallel_text_dataset  # noqa: F401
