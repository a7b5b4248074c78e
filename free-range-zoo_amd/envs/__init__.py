"""Environment entry points, named as in the reference (``free_range_zoo/envs/__init__.py``)."""
from free_range_zoo_amd.envs import wildfire_v0
from free_range_zoo_amd.envs import cybersecurity_v0
from free_range_zoo_amd.envs import rideshare_v0

__all__ = ['wildfire_v0', 'cybersecurity_v0', 'rideshare_v0']
