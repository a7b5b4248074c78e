from free_range_zoo_amd.envs.wildfire.env.wildfire import raw_env, env, parallel_env
