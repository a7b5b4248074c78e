"""Scripted cybersecurity baselines (mirrors free_range_zoo/envs/cybersecurity/baselines): device-side, one launch per decision."""
from free_range_zoo_amd.envs.cybersecurity.baselines.camp import CampDefenderBaseline  # noqa: F401
from free_range_zoo_amd.envs.cybersecurity.baselines.noop import NoopBaseline  # noqa: F401
from free_range_zoo_amd.envs.cybersecurity.baselines.random import RandomBaseline  # noqa: F401
from free_range_zoo_amd.envs.cybersecurity.baselines.patched import PatchedAttackerBaseline, PatchedDefenderBaseline  # noqa: F401
from free_range_zoo_amd.envs.cybersecurity.baselines.exploited import ExploitedAttackerBaseline, ExploitedDefenderBaseline  # noqa: F401
