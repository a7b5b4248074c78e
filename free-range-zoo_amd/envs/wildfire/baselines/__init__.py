"""Scripted wildfire baselines (mirrors free_range_zoo/envs/wildfire/baselines): device-side, one launch per decision."""
from free_range_zoo_amd.envs.wildfire.baselines.noop import NoopBaseline  # noqa: F401
from free_range_zoo_amd.envs.wildfire.baselines.random import RandomBaseline  # noqa: F401
from free_range_zoo_amd.envs.wildfire.baselines.strongest import StrongestBaseline  # noqa: F401
from free_range_zoo_amd.envs.wildfire.baselines.weakest import WeakestBaseline  # noqa: F401
