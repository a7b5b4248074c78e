"""Agent that always acts on the first arrived passenger and works on a task until completion. (free_range_zoo/envs/rideshare/baselines/fifo_Tfocus.py:9-80)."""
from free_range_zoo_amd.envs.rideshare.baselines._task_policy import TaskPolicyBaseline


class FirstInFirstOutTfocusBaseline(TaskPolicyBaseline):
    """Agent that always acts on the first arrived passenger and works on a task until completion."""
    kind = 'fifo_focus'
