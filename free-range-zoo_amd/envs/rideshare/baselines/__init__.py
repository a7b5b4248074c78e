"""Scripted rideshare baselines (mirrors free_range_zoo/envs/rideshare/baselines): device-side, one launch per decision."""
from free_range_zoo_amd.envs.rideshare.baselines.noop import NoopBaseline  # noqa: F401
from free_range_zoo_amd.envs.rideshare.baselines.random import RandomBaseline  # noqa: F401
from free_range_zoo_amd.envs.rideshare.baselines.fifo_Tfocus import FirstInFirstOutTfocusBaseline  # noqa: F401
from free_range_zoo_amd.envs.rideshare.baselines.fifo_Tglobal import FirstInFirstOutTglobalBaseline  # noqa: F401
from free_range_zoo_amd.envs.rideshare.baselines.greedy_Tfocus import GreedyTaskFocus  # noqa: F401
from free_range_zoo_amd.envs.rideshare.baselines.greedy_Tglobal import GreedyTaskGlobal  # noqa: F401
