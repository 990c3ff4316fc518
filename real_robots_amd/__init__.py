"""real_robots_amd: MI355X-native batched implementation of the REALRobot env.step() hot path."""
__version__ = '0.1.0'
