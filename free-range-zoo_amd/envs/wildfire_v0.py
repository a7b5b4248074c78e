"""``from free_range_zoo_amd.envs import wildfire_v0`` — same entry points as the reference's ``envs/wildfire_v0.py``."""
from free_range_zoo_amd.envs.wildfire.env.wildfire import raw_env, env, parallel_env

__all__ = ['raw_env', 'env', 'parallel_env']
