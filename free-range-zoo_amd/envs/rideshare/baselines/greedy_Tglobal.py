"""Agent that acts on the soonest-to-completion passenger, re-deciding among all passengers every step. (free_range_zoo/envs/rideshare/baselines/greedy_Tglobal.py:10-99)."""
from free_range_zoo_amd.envs.rideshare.baselines._task_policy import TaskPolicyBaseline


class GreedyTaskGlobal(TaskPolicyBaseline):
    """Agent that acts on the soonest-to-completion passenger, re-deciding among all passengers every step."""
    kind = 'greedy_global'
