"""Agent that always fights the strongest available fire (free_range_zoo/envs/wildfire/baselines/strongest.py:8-62)."""
from free_range_zoo_amd.envs.wildfire.baselines._extreme import ExtremeFireBaseline


class StrongestBaseline(ExtremeFireBaseline):
    """Agent that always fights the strongest available fire."""
    weakest = False
