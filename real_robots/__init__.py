"""Compatibility alias: `import real_robots` resolves to the MI355X-native package `real_robots_amd`, so existing agents
(`from real_robots.policy import BasePolicy`, `real_robots.evaluate(...)`) and goal datasets pickled against
`real_robots.envs.env.Goal` work unchanged."""
import sys

import real_robots_amd as _impl
from real_robots_amd import *  # noqa: F401,F403
from real_robots_amd import envs, evaluate, policy, registry, spaces  # noqa: F401
from real_robots_amd.envs import env as _env, robot as _robot

__version__ = _impl.__version__
getPackageDataPath = _impl.getPackageDataPath
make = _impl.make
for _name, _mod in (('envs', envs), ('envs.env', _env), ('envs.robot', _robot), ('policy', policy),
                    ('registry', registry), ('spaces', spaces)):
    sys.modules[__name__ + '.' + _name] = _mod
sys.modules[__name__ + '.evaluate'] = sys.modules['real_robots_amd.evaluate']
evaluate = _impl.evaluate
# Goal datasets written here must load in the reference package (and anywhere `real_robots_amd` is not installed): the
# class pickles under the reference's path, which this alias serves (env.py:15-24, generate_goals.py:435-436).
_env.Goal.__module__ = __name__ + '.envs.env'
