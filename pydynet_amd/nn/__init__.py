from .modules import *  # noqa: F401,F403
from .parameter import Parameter  # noqa: F401
from . import init, functional  # noqa: F401
