"""Task base + registry (neurst/tasks/task.py:27-111)."""
from neurst_amd.utils.registry import setup_registry


class Task(object):
    REGISTRY_NAME = "task"

    def __init__(self, args):
        self._args = args

    @staticmethod
    def class_or_method_args():
        return []

    def get_config(self):
        return {}

    def build_model(self, args, name=None, **kwargs):
        raise NotImplementedError

    def example_to_input(self, batch_of_data, mode):
        raise NotImplementedError

    def create_and_batch_dataset(self, dataset, mode, args, rank=0, world=1):
        raise NotImplementedError


build_task, register_task = setup_registry(Task.REGISTRY_NAME, base_class=Task, backend="pt")
