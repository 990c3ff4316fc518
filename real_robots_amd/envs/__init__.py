from .env import REALRobotEnv, Goal, EnvCamera  # noqa: F401
from .robot import Kuka  # noqa: F401
