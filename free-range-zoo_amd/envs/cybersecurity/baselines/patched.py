"""Agents that pick the most patched node and attack / patch it, repositioning every three timesteps
(free_range_zoo/envs/cybersecurity/baselines/patched.py:9-152)."""
from free_range_zoo_amd.envs.cybersecurity.baselines._focus import FocusPolicyBaseline


class PatchedAttackerBaseline(FocusPolicyBaseline):
    """Agent that picks the most patched node and attacks it for three consecutive steps."""
    kind = 'patched_attacker'


class PatchedDefenderBaseline(FocusPolicyBaseline):
    """Agent that picks the most patched node and patches it for three consecutive steps."""
    kind = 'patched_defender'
