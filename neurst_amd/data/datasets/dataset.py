"""Dataset base + registry (neurst/data/datasets/dataset.py:27-84)."""
from neurst_amd.utils.registry import setup_registry


class Dataset(object):
    REGISTRY_NAME = "dataset"

    def __init__(self):
        pass

    @staticmethod
    def class_or_method_args():
        return []

    @property
    def status(self):
        raise NotImplementedError

    def build_iterator(self, map_func=None, shard_id=0, total_shards=1):
        raise NotImplementedError


build_dataset, register_dataset = setup_registry(Dataset.REGISTRY_NAME, base_class=Dataset, backend="pt")
