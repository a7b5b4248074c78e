from free_range_zoo_amd.wrappers.action_task import action_mapping_wrapper_v0  # noqa: F401
from free_range_zoo_amd.wrappers.space_validator import space_validator_wrapper_v0  # noqa: F401
