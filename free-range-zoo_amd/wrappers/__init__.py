from free_range_zoo_amd.wrappers.action_task import action_mapping_wrapper_v0  # noqa: F401
