"""``from free_range_zoo_amd.envs import cybersecurity_v0`` — same entry points as the reference's ``envs/cybersecurity_v0.py``."""
from free_range_zoo_amd.envs.cybersecurity.env.cybersecurity import raw_env, env, parallel_env

__all__ = ['raw_env', 'env', 'parallel_env']
