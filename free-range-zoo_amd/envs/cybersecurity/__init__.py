from free_range_zoo_amd.envs.cybersecurity.env.cybersecurity import raw_env, env, parallel_env
