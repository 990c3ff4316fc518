"""Environment registry: the 18 ids `REALRobot2020-{R1,R2}{J,C,M}{1,2,3}-v0` of the reference
(real_robots/__init__.py:16-28). Registers with gym when it is importable; `make()` works either way."""
_REGISTRY = {}


def register(id, entry_point, kwargs):
    _REGISTRY[id] = (entry_point, dict(kwargs))
    try:  # pragma: no cover
        from gym.envs.registration import register as gym_register
        gym_register(id=id, entry_point='real_robots_amd.envs:REALRobotEnv', kwargs=kwargs)
    except Exception:
        pass


def make(id, **overrides):
    if id not in _REGISTRY:
        raise KeyError("No registered env with id: %s" % id)
    entry_point, kwargs = _REGISTRY[id]
    kw = dict(kwargs)
    kw.update(overrides)
    return entry_point(**kw)


def registered_ids():
    return sorted(_REGISTRY)
