from free_range_zoo_amd.envs.rideshare.env.rideshare import raw_env, env, parallel_env
