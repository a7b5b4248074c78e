"""Agent that camps on one node and patches it (free_range_zoo/envs/cybersecurity/baselines/camp.py:9-60)."""
from free_range_zoo_amd.envs.cybersecurity.baselines._focus import FocusPolicyBaseline


class CampDefenderBaseline(FocusPolicyBaseline):
    """Moves to node ``agent index % nodes`` and patches it forever.

    The reference reads the mapping under the key ``'action_task_mappings'`` (camp.py:40), which its own wrapper does not
    provide; both that key and the wrapper's ``'agent_action_mapping'`` are accepted here.
    """
    kind = 'camp_defender'
    mapping_key = 'action_task_mappings'

    def __init__(self, *args, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        self.agent_index = int(self.agent_name.split('_')[-1])

    def _camp_target(self, nodes: int) -> int:
        return self.agent_index % nodes
