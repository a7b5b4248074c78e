"""Agent that always acts on the first arrived passenger, re-deciding among all passengers every step. (free_range_zoo/envs/rideshare/baselines/fifo_Tglobal.py:9-74)."""
from free_range_zoo_amd.envs.rideshare.baselines._task_policy import TaskPolicyBaseline


class FirstInFirstOutTglobalBaseline(TaskPolicyBaseline):
    """Agent that always acts on the first arrived passenger, re-deciding among all passengers every step."""
    kind = 'fifo_global'
