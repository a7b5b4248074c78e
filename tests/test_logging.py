"""Logging tap: the CSV files written along the golden trajectories are the reference's.

tests/golden/logs_csv.npz holds the text of the per-env CSV files the unmodified reference wrote (utils/logging_handlers.py:36-114)
along one golden trajectory per domain (tools/refharness/make_golden.py logs).  The GPU tests replay the same actions and injected
randomness through the HIP envs with ``log_directory`` set and compare the files cell by cell; the CPU tests cover the logger itself
(row format, reset semantics, asynchronous hand-off, failure reporting) on host tensors.
"""
import csv
import io
import os

import numpy as np
import pytest
import torch

import configs
import golden_util as G


def parse(text):
    return list(csv.reader(io.StringIO(text)))


def assert_csv_equal(got: str, want: str, what: str):
    """Identical text, or — cell by cell — identical strings / floats within 1e-5 (rewards may differ in the last ulp)."""
    if got == want:
        return
    g, w = parse(got), parse(want)
    assert len(g) == len(w), f'{what}: {len(g)} rows, reference wrote {len(w)}'
    assert g[0] == w[0], f'{what}: header'
    for r, (gr, wr) in enumerate(zip(g[1:], w[1:])):
        assert len(gr) == len(wr), f'{what} row {r}'
        for name, gc, wc in zip(g[0], gr, wr):
            if gc == wc:
                continue
            try:
                ok = name.endswith('_rewards') and abs(float(gc) - float(wc)) <= 1e-5 * max(1.0, abs(float(wc)))
            except ValueError:
                ok = False
            assert ok, f'{what} row {r} column {name}: {gc!r} != {wc!r}'


# ------------------------------------------------------------------------------------------------------------------
# the logger on host tensors
# ------------------------------------------------------------------------------------------------------------------
def _toy_state(B, value):
    from free_range_zoo_amd.envs.cybersecurity.env.structures.state import CybersecurityState
    return CybersecurityState(network_state=torch.full((B, 3), value, dtype=torch.int32), location=torch.zeros((B, 1), dtype=torch.int32),
                              presence=torch.ones((B, 2), dtype=torch.bool))


@pytest.mark.parametrize('asynchronous', [False, True])
def test_csv_logger_rows_and_reset(tmp_path, asynchronous):
    from free_range_zoo_amd.utils.logging_handlers import CSVLogger
    B, agents = 2, ['attacker_1', 'defender_1']
    log = CSVLogger(str(tmp_path / 'logs'), B, asynchronous=asynchronous)
    maps = {a: torch.nested.nested_tensor([torch.tensor([0, 2]), torch.tensor([], dtype=torch.int64)], layout=torch.jagged) for a in agents}
    obs_maps = {a: torch.arange(3).repeat(B, 1, 1) for a in agents}
    log.log_environment(_toy_state(B, 1), None, None, maps, obs_maps, None, None, 'demo', agents, reset=True)
    actions = {a: torch.tensor([[1, 0], [2, -1]], dtype=torch.int32) for a in agents}
    rewards = {a: torch.tensor([0.5, -1.25]) for a in agents}
    log.log_environment(_toy_state(B, 2), actions, rewards, maps, obs_maps, torch.tensor([1, 1]), torch.tensor([False, True]), 'demo', agents,
                        extra={'adj': ['x', 'y']})
    log.flush()
    rows = parse(open(tmp_path / 'logs' / '1.csv').read())
    assert rows[0][:3] == ['network_state', 'location', 'presence'] and rows[0][-1] == 'description'
    assert rows[1][rows[0].index('step')] == '-1' and rows[1][rows[0].index('attacker_1_action')] == 'NULL'
    assert rows[2][rows[0].index('attacker_1_action')] == '[2, -1]' and rows[2][rows[0].index('defender_1_rewards')] == '-1.25'
    assert rows[2][rows[0].index('complete')] == 'True' and rows[2][rows[0].index('attacker_1_action_map')] == '[]'
    assert rows[2][rows[0].index('attacker_1_observation_map')] == '[[0, 1, 2]]' and rows[2][0] == '[2, 2, 2]'
    # a reset starts the files again (utils/env.py:140-143 + logging_handlers.py:74-75)
    log.reset()
    log.log_environment(_toy_state(B, 3), None, None, maps, obs_maps, None, None, None, agents, reset=True)
    log.close()
    rows = parse(open(tmp_path / 'logs' / '0.csv').read())
    assert len(rows) == 2 and rows[1][0] == '[3, 3, 3]' and rows[1][-1] == 'NULL'
    with pytest.raises(RuntimeError):  # agents must not change between resets (logging_handlers.py:66-72)
        log.log_environment(_toy_state(B, 3), actions, rewards, maps, obs_maps, torch.tensor([1, 1]), torch.tensor([False, True]), None,
                            agents[:1])


def test_csv_logger_refuses_a_used_directory_and_reports_writer_failures(tmp_path):
    from free_range_zoo_amd.utils.logging_handlers import CSVLogger
    used = tmp_path / 'used'
    used.mkdir()
    (used / 'x').write_text('x')
    with pytest.raises(FileExistsError):
        CSVLogger(str(used), 1)
    CSVLogger(str(used), 1, override_initialization_check=True)
    log = CSVLogger(str(tmp_path / 'gone'), 1)
    os.rmdir(tmp_path / 'gone')  # the writer thread cannot create its files any more
    maps = {'a_1': torch.zeros((1, 1), dtype=torch.int64)}
    log.log_environment(_toy_state(1, 1), None, None, maps, maps, None, None, None, ['a_1'], reset=True)
    with pytest.raises(RuntimeError):
        log.flush()


SQL_TABLES = ('simulation', 'environment', 'agent', 'environment_timestep', 'wildfire_environment_log', 'rideshare_environment_log',
              'cybersecurity_environment_log', 'agent_log')


def dump_sqlite(path):
    """Same dump as tools/refharness/make_golden.py made of the reference's database: every table in id order, without the run date."""
    import sqlite3
    con = sqlite3.connect(path)
    out = {}
    for table in SQL_TABLES:
        cur = con.execute(f'SELECT * FROM {table} ORDER BY id')
        columns = [c[0] for c in cur.description]
        rows = [list(r) for r in cur.fetchall()]
        if table == 'simulation':
            keep = [i for i, c in enumerate(columns) if c != 'timestamp']
            columns, rows = [columns[i] for i in keep], [[r[i] for i in keep] for r in rows]
        out[table] = {'columns': columns, 'rows': rows}
    con.close()
    return out


@pytest.mark.parametrize('asynchronous', [False, True])
def test_sql_logger_tables(tmp_path, asynchronous):
    """SQLLogger on host tensors: the schema of the reference (utils/sql_logging.py:12-118) and its per-table row order
    (logging_handlers.py:132-241)."""
    from free_range_zoo_amd.utils.logging_handlers import SQLLogger
    B, agents = 2, ['attacker_1', 'defender_1']
    path = tmp_path / 'log.db'
    log = SQLLogger(f'sqlite:///{path}', 'cybersecurity_v0', B, asynchronous=asynchronous)
    maps = {a: torch.nested.nested_tensor([torch.tensor([0, 2]), torch.tensor([], dtype=torch.int64)], layout=torch.jagged) for a in agents}
    obs_maps = {a: torch.arange(3).repeat(B, 1, 1) for a in agents}
    extra = {'adj_matrix': ['[[0, 1], [1, 0]]'] * B}
    with pytest.raises(RuntimeError):  # logging_handlers.py:175-176
        log.log_environment(_toy_state(B, 1), None, None, maps, obs_maps, torch.zeros(B, dtype=torch.int32), None, None, agents, extra=extra, reset=True)
    log.reset(log_label='demo', log_description='two envs', agents=agents)
    log.log_environment(_toy_state(B, 1), None, None, maps, obs_maps, torch.zeros(B, dtype=torch.int32), None, None, agents, extra=extra, reset=True)
    later = agents + ['defender_2']
    maps['defender_2'], obs_maps['defender_2'] = maps['defender_1'], obs_maps['defender_1']
    actions = {a: torch.tensor([[1, 0], [2, -1]], dtype=torch.int32) for a in later}
    rewards = {a: torch.tensor([0.5, -1.25]) for a in later}
    log.log_environment(_toy_state(B, 2), actions, rewards, maps, obs_maps, torch.tensor([1, 1]), torch.tensor([False, True]), None, later,
                        extra=extra)
    log.close()
    t = dump_sqlite(str(path))
    assert t['simulation'] == {'columns': ['id', 'name', 'description'], 'rows': [[1, 'demo', 'two envs']]}
    assert t['environment']['rows'] == [[1, 1, 0], [2, 1, 1]]
    # agent-major at reset; an agent that appears later joins env by env
    assert t['agent']['rows'] == [[1, 1, 'attacker_1'], [2, 2, 'attacker_1'], [3, 1, 'defender_1'], [4, 2, 'defender_1'], [5, 1, 'defender_2'],
                                  [6, 2, 'defender_2']]
    assert t['environment_timestep']['columns'] == ['environment_id', 'id', 'timestep']
    assert t['environment_timestep']['rows'] == [[1, 1, 0], [2, 2, 0], [1, 3, 1], [2, 4, 1]]
    assert t['cybersecurity_environment_log']['rows'][2] == [3, 3, '[2, 2, 2]', '[0]', '[True, True]', '[[0, 1], [1, 0]]']
    assert t['wildfire_environment_log']['rows'] == [] and t['rideshare_environment_log']['rows'] == []
    # no agent rows for the reset log; rewards are stored truncated (logging_handlers.py:228), env-major then agent
    assert t['agent_log']['rows'][0] == [1, 3, 1, 0, 0, 1, '[0, 2]', '[[0, 1, 2]]']
    assert t['agent_log']['rows'][3] == [4, 4, 2, -1, -1, 2, '[]', '[[0, 1, 2]]']
    assert [r[2] for r in t['agent_log']['rows']] == [1, 3, 5, 2, 4, 6]


def test_sql_logger_rejects_unknown_domains(tmp_path):
    from free_range_zoo_amd.utils.logging_handlers import SQLLogger
    with pytest.raises(NotImplementedError):
        SQLLogger(f'sqlite:///{tmp_path / "x.db"}', 'chess_v0', 2)


# ------------------------------------------------------------------------------------------------------------------
# the envs' files against the reference's
# ------------------------------------------------------------------------------------------------------------------
def _replay(domain, tmp_path, stacked=False, sql=False):
    logs = np.load(G.golden_path('logs_csv.npz'))
    name = str(logs[f'{domain}_name'])
    data = np.load(G.golden_path(f'traj_{domain}_{name}.npz'))
    directory = str(tmp_path / domain)
    target = f'sqlite:///{directory}.db' if sql else directory
    if domain == 'wildfire':
        from test_hip_wildfire import make_env
        from free_range_zoo_amd import _capi
        build, kwargs = configs.WILDFIRE_GOLDEN[name]
        cfg = G.load_cfg(data, _capi.frz_wildfire_cfg)
        shapes = [(3, cfg.parallel_envs, cfg.grid_height * cfg.grid_width), (5, cfg.parallel_envs, cfg.num_agents)]
        keys = ('field_randomness', 'agent_randomness')
    elif domain == 'cybersecurity':
        from test_hip_cybersecurity import make_env
        from free_range_zoo_amd import _capi
        build, kwargs = configs.CYBER_GOLDEN[name]
        cfg = G.load_cfg(data, _capi.frz_cybersecurity_cfg)
        shapes = [(1, cfg.parallel_envs, cfg.num_nodes), (1, cfg.parallel_envs, cfg.num_attackers + cfg.num_defenders)]
        keys = ('network_randomness', 'agent_randomness')
    else:
        from test_hip_rideshare import make_env
        from free_range_zoo_amd import _capi
        build, kwargs = configs.RIDESHARE_GOLDEN[name], {}
        cfg = G.load_cfg(data, _capi.frz_rideshare_cfg)
        shapes = keys = None
    B = cfg.parallel_envs
    env = make_env(build, B, None if cfg.max_steps < 0 else cfg.max_steps, log_directory=target, **kwargs)
    env.reset(seed=torch.arange(B, dtype=torch.int32), options={'log_description': f'golden {name}', 'log_label': f'run {name}'})
    for t in range(int(data['steps'])):
        p = f's{t}_'
        actions = {agent: torch.from_numpy(data[p + 'actions'][a]).cuda() for a, agent in enumerate(env.agents)}
        if stacked:
            actions = torch.stack([actions[agent] for agent in env.agents]).contiguous()
        if keys is None:
            env.step(actions)
        else:
            stepped = bool(data[p + 'stepped'])
            rnd = tuple(torch.from_numpy(data[p + k]) if stepped else torch.zeros(s) for k, s in zip(keys, shapes))
            env.step(actions, randomness=rnd)
    env.check()
    env.close()
    if sql:
        return f'{directory}.db'
    want = logs[domain]
    assert sorted(os.listdir(directory)) == sorted(f'{i}.csv' for i in range(B))
    for i in range(B):
        assert_csv_equal(open(os.path.join(directory, f'{i}.csv')).read(), str(want[i]), f'{domain} {name} env {i}')
    return directory


@pytest.mark.gpu
@pytest.mark.parametrize('domain', ['wildfire', 'cybersecurity', 'rideshare'])
def test_env_logs_equal_the_reference_files(domain, tmp_path):
    _replay(domain, tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize('domain', ['wildfire', 'cybersecurity', 'rideshare'])
def test_env_sql_logs_equal_the_reference_database(domain, tmp_path):
    """log_directory='sqlite:///...': table by table the database the reference's SQLLogger wrote along the same trajectory
    (tests/golden/logs_sql.npz, tools/refharness/make_golden.py logs_sql)."""
    import json
    path = _replay(domain, tmp_path, sql=True)
    want = json.loads(str(np.load(G.golden_path('logs_sql.npz'))[domain]))
    got = dump_sqlite(path)
    for table in SQL_TABLES:
        assert got[table]['columns'] == want[table]['columns'], table
        assert len(got[table]['rows']) == len(want[table]['rows']), f'{domain} {table}: {len(got[table]["rows"])} rows, reference wrote {len(want[table]["rows"])}'
        for r, (g, w) in enumerate(zip(got[table]['rows'], want[table]['rows'])):
            assert g == w, f'{domain} {table} row {r}: {g} != {w}'


@pytest.mark.gpu
def test_stacked_actions_are_logged_too(tmp_path):
    _replay('cybersecurity', tmp_path, stacked=True)


@pytest.mark.gpu
def test_the_reference_quickstart_runs_with_the_package_renamed(tmp_path):
    """docs/source/introduction/quickstart.md of the reference, full script, with `free_range_zoo` -> `free_range_zoo_amd` and
    the device set to the GPU: configuration classes, env creation with logging, the action-task wrapper, the baselines and the
    rollout loop are used exactly as documented there."""
    from free_range_zoo_amd.envs.rideshare.env.structures.configuration import (RewardConfiguration, PassengerConfiguration, AgentConfiguration,
                                                                               RideshareConfiguration)
    from free_range_zoo_amd.envs import rideshare_v0
    from free_range_zoo_amd.wrappers.action_task import action_mapping_wrapper_v0

    reward_config = RewardConfiguration(pick_cost=-0.1, move_cost=-0.02, drop_cost=0.0, noop_cost=-0.1, accept_cost=-0.1, pool_limit_cost=-4.0,
                                        use_pooling_rewards=False, use_variable_move_cost=False, use_waiting_costs=False,
                                        wait_limit=torch.tensor([5, 3, 3], dtype=torch.int32), long_wait_time=5, general_wait_cost=-0.1,
                                        long_wait_cost=-0.5)
    passenger_config = PassengerConfiguration(schedule=torch.tensor([[0, 0, 0, 0, 1, 1, 3], ], dtype=torch.int32))
    agent_config = AgentConfiguration(start_positions=torch.tensor([[1, 1], [2, 2]]), pool_limit=2, use_diagonal_travel=False, use_fast_travel=True)
    rideshare_config = RideshareConfiguration(grid_height=5, grid_width=5, reward_config=reward_config, passenger_config=passenger_config,
                                              agent_config=agent_config)
    log_directory = str(tmp_path / 'test_logging')
    env = rideshare_v0.parallel_env(max_steps=100, parallel_envs=1, configuration=rideshare_config, device=torch.device('cuda'),
                                    log_directory=log_directory)
    env.reset()
    env = action_mapping_wrapper_v0(env)
    observations, infos = env.reset()

    from free_range_zoo_amd.envs.rideshare.baselines import NoopBaseline, RandomBaseline
    agents = {env.agents[0]: NoopBaseline(agent_name='agent_0', parallel_envs=1), env.agents[1]: RandomBaseline(agent_name='agent_1', parallel_envs=1)}
    steps = 0
    while not torch.all(env.finished):
        for agent_name, agent in agents.items():
            agent.observe(observations[agent_name][0])
        agent_actions = {agent_name: agents[agent_name].act(action_space=env.action_space(agent_name)) for agent_name in env.agents}
        observations, rewards, terminations, truncations, infos = env.step(agent_actions)
        steps += 1
    env.close()
    assert steps == 100
    rows = parse(open(os.path.join(log_directory, '0.csv')).read())
    assert len(rows) == 1 + 1 + 100 and rows[0][:2] == ['agents', 'passengers'] and rows[-1][rows[0].index('complete')] == 'True'
    assert rows[1][rows[0].index('passengers')] == '[[0, 0, 0, 1, 1, 3, 0, -1, 0, -1, -1]]'
