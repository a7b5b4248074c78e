"""Agent that always fights the weakest available fire (free_range_zoo/envs/wildfire/baselines/weakest.py:8-64)."""
from free_range_zoo_amd.envs.wildfire.baselines._extreme import ExtremeFireBaseline


class WeakestBaseline(ExtremeFireBaseline):
    """Agent that always fights the weakest available fire."""
    weakest = True
