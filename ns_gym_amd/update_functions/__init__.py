from .distribution import *  # noqa: F401,F403
from .single_param import *  # noqa: F401,F403
