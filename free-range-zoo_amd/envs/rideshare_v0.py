"""``from free_range_zoo_amd.envs import rideshare_v0`` — same entry points as the reference's ``envs/rideshare_v0.py``."""
from free_range_zoo_amd.envs.rideshare.env.rideshare import raw_env, env, parallel_env

__all__ = ['raw_env', 'env', 'parallel_env']
