"""real_robots_amd: MI355X-native batched implementation of the REALRobot env.step() hot path.

Drop-in surface of the reference package `real_robots` (v0.1.21): `make("REALRobot2020-R2J3-v0")`, `BasePolicy`,
`evaluate(...)`, `envs.REALRobotEnv/Goal/EnvCamera`; plus `BatchedREALRobotEnv` for thousands of envs per GPU.
"""
__version__ = '0.1.0'

import os

from .registry import register, make, registered_ids  # noqa: F401
from .policy import BasePolicy, BatchedPolicy  # noqa: F401
from .batched import BatchedREALRobotEnv  # noqa: F401
from .envs import REALRobotEnv  # noqa: F401
from .evaluate import evaluate, evaluate_batched  # noqa: F401

# the 18 environment ids of the reference (real_robots/__init__.py:16-28)
for _n_obj in [1, 2, 3]:
    for _obs, _rnd in zip([True, False], ["R1", "R2"]):
        for _action_type in ['joints', 'cartesian', 'macro_action']:
            register(id='REALRobot2020-{}{}{}-v0'.format(_rnd, _action_type[0].upper(), _n_obj),
                     entry_point=REALRobotEnv,
                     kwargs={'additional_obs': _obs, 'objects': _n_obj, 'action_type': _action_type})


def getPackageDataPath():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
