"""Agent that acts on the soonest-to-completion passenger and works on a task until completion. (free_range_zoo/envs/rideshare/baselines/greedy_Tfocus.py:10-115)."""
from free_range_zoo_amd.envs.rideshare.baselines._task_policy import TaskPolicyBaseline


class GreedyTaskFocus(TaskPolicyBaseline):
    """Agent that acts on the soonest-to-completion passenger and works on a task until completion."""
    kind = 'greedy_focus'
