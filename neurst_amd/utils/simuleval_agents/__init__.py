"""Agent registry (neurst/utils/simuleval_agents/__init__.py): `register_agent` keeps SimulEval agents by snake-case name."""
import re

AGENTS = {}


def register_agent(cls):
    name = re.sub(r"(?<!^)(?=[A-Z])", "_", cls.__name__).lower()
    AGENTS[name] = cls
    AGENTS[cls.__name__] = cls
    return cls


def build_agent(name, args):
    from neurst_amd.utils.simuleval_agents import simul_trans_text_agent  # noqa: F401  (registers)
    return AGENTS[name](args)
