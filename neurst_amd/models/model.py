"""BaseModel + model registry (neurst/models/model.py, neurst/models/__init__.py).  Models are built through the
class method ``new`` (create_fn="new"), exactly like the reference registry does."""
from neurst_amd.utils.registry import setup_registry


class BaseModel(object):
    REGISTRY_NAME = "model"

    def __init__(self, args, name=None):
        self._args = args
        self.name = name

    @property
    def args(self):
        return self._args

    @staticmethod
    def class_or_method_args():
        return []

    @classmethod
    def new(cls, args, *extra, **kwargs):
        raise NotImplementedError

    @classmethod
    def build_model_args_by_name(cls, name):
        return None


build_model, register_model = setup_registry(BaseModel.REGISTRY_NAME, base_class=BaseModel, create_fn="new",
                                             backend="pt")
