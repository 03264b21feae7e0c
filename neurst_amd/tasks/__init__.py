from neurst_amd.tasks.task import Task, build_task, register_task  # noqa: F401
from neurst_amd.tasks import speech2text  # noqa: F401
from neurst_amd.tasks import seq2seq  # noqa: F401
