"""`python -m real_robots_amd.cli`: the reference's `real-robots-demo` smoke run (real_robots/cli.py:12-64) — a random
policy for a few hundred steps on REALRobot2020-R2J3-v0 — on the HIP path (headless)."""
import argparse
import time

import numpy as np

from . import make
from .policy import BasePolicy


class RandomPolicy(BasePolicy):
    """README "Usage" policy: resample the joint command with probability 0.05 per step, render every step."""

    def __init__(self, action_space, observation_space):
        super().__init__(action_space, observation_space)
        self.action = action_space.sample()

    def step(self, observation, reward, done):
        if np.random.rand() < 0.05:
            self.action = self.action_space.sample()
        return self.action


def run_episode(env, pi, steps):
    observation = env.reset()
    reward, done = 0, False
    for _ in range(steps):
        observation, reward, done, _ = env.step(pi.step(observation, reward, done))
    return observation


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--env', default='REALRobot2020-R2J3-v0')
    a = ap.parse_args(argv)
    env = make(a.env)
    pi = RandomPolicy(env.action_space, env.observation_space)
    t0 = time.time()
    obs = run_episode(env, pi, a.steps)
    print("ran %d steps of %s in %.2f s; joints %s" % (a.steps, a.env, time.time() - t0, np.round(obs['joint_positions'], 3)))


if __name__ == '__main__':
    main()
