from .env import REALRobotEnv, Goal, EnvCamera, EyeCamera  # noqa: F401
from .robot import Kuka  # noqa: F401
