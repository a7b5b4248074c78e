"""Logging tap for the batched environments (mirrors free_range_zoo/utils/logging_handlers.py:21-114).

Same interface and the same files as the reference's ``CSVLogger`` — one ``<env>.csv`` per parallel environment, one row per
reset / step, identical columns and cell text — but taken off the step path: ``log_environment`` only enqueues device-to-host
copies of what a row needs on the env's stream (pinned destinations, no host synchronisation) followed by an event; a writer thread
waits for the event, formats the rows with pandas exactly as the reference does and appends them.  The step loop never waits for
the file system; ``flush()`` / ``close()`` drain the queue.

``SQLLogger`` (logging_handlers.py:117-241: sqlite / postgres through SQLAlchemy, schema of utils/sql_logging.py) rides the same tap:
the rows of a step are inserted by the writer thread, table by table, in one transaction.
"""
import datetime
import os
import queue
import threading
from typing import Any, Dict, List, Optional

import pandas as pd
import torch


class Logger:
    """Abstract logger interface (logging_handlers.py:21-33)."""

    def log_environment(self, *args, **kwargs):
        raise NotImplementedError

    def log_agent(self, *args, **kwargs):
        raise NotImplementedError

    def reset(self, *args, **kwargs):
        pass

    def flush(self) -> None:
        pass

    def close(self) -> None:
        pass


def _to_host(value: torch.Tensor) -> torch.Tensor:
    """Stream-ordered copy to pinned host memory (the result is valid once the event recorded after it has completed)."""
    if not isinstance(value, torch.Tensor) or value.device.type == 'cpu':
        return value
    return value.detach().to('cpu', non_blocking=True)


def _snapshot_mapping(mapping) -> Any:
    """A per-env list-like (jagged nested tensor, with or without explicit lengths, or a dense [B, ...] tensor) -> host parts."""
    if isinstance(mapping, torch.Tensor) and mapping.is_nested:
        lengths = mapping.lengths()
        return ('jagged', _to_host(mapping.values()), _to_host(mapping.offsets()), None if lengths is None else _to_host(lengths))
    return ('dense', _to_host(mapping))


def _mapping_rows(parts) -> List[str]:
    """``[str(mapping.tolist()) for mapping in nested]`` (logging_handlers.py:92-93) from the host parts."""
    if parts[0] == 'dense':
        return [str(row.tolist()) for row in parts[1]]
    _, values, offsets, lengths = parts
    offsets = offsets.tolist()
    counts = [offsets[i + 1] - offsets[i] for i in range(len(offsets) - 1)] if lengths is None else lengths.tolist()
    return [str(values[start:start + count].tolist()) for start, count in zip(offsets, counts)]


class _TapLogger(Logger):
    """The asynchronous hand-over shared by the loggers: ``_submit(job)`` from the step loop, ``_handle(job)`` on the writer thread
    (``asynchronous=False``: inline).  A failure of the writer is raised by the next ``log_environment`` / ``flush`` / ``close``."""

    _thread_name = 'frz-logger'

    def __init__(self, asynchronous: bool = True):
        self._asynchronous = asynchronous
        self._jobs: 'queue.Queue' = queue.Queue()
        self._failure: Optional[BaseException] = None
        self._worker: Optional[threading.Thread] = None

    def flush(self) -> None:
        """Block until every row handed over so far is written."""
        if self._worker is not None:
            self._jobs.join()
        self._raise_failure()

    def close(self) -> None:
        if self._worker is not None:
            self._jobs.join()
            self._jobs.put(None)
            self._worker.join()
            self._worker = None
        self._raise_failure()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _raise_failure(self) -> None:
        if self._failure is not None:
            failure, self._failure = self._failure, None
            raise RuntimeError('the log writer thread failed') from failure

    def _submit(self, job) -> None:
        if not self._asynchronous:
            self._handle(job)
            return
        if self._worker is None:
            self._worker = threading.Thread(target=self._drain, name=self._thread_name, daemon=True)
            self._worker.start()
        self._jobs.put(job)

    def _drain(self) -> None:
        while True:
            job = self._jobs.get()
            try:
                if job is None:
                    return
                if self._failure is None:
                    self._handle(job)
            except BaseException as failure:  # surfaced by the next log_environment / flush / close
                self._failure = failure
            finally:
                self._jobs.task_done()

    @staticmethod
    def _completion_event():
        if torch.cuda.is_available() and torch.cuda.is_initialized():
            event = torch.cuda.Event()
            event.record()
            return event
        return None

    def _handle(self, job) -> None:
        raise NotImplementedError


class CSVLogger(_TapLogger):
    """CSV logger writing the reference's format (logging_handlers.py:36-114) from a background writer thread."""

    _thread_name = 'frz-csv-logger'

    def __init__(self, log_directory: str, parallel_envs: int, override_initialization_check: bool = False, asynchronous: bool = True):
        super().__init__(asynchronous)
        self.log_directory = log_directory
        self.parallel_envs = parallel_envs
        self._are_logs_initialized = False
        self._agent_set = None
        if not override_initialization_check and os.path.exists(log_directory):
            if os.listdir(log_directory):
                raise FileExistsError('The logging output directory already exists. Set override_initialization_check or rename.')
        if not os.path.exists(log_directory):
            os.mkdir(log_directory)

    # ------------------------------------------------------------------------------------------- producer side
    def log_environment(self,
                        state,
                        actions,
                        rewards,
                        agent_action_mapping,
                        agent_observation_mapping,
                        num_moves,
                        finished,
                        log_description,
                        agents,
                        extra=None,
                        reset=False):
        """Same arguments as the reference.  ``state`` needs ``to_dataframe_parts()`` (utils/state.py); ``extra`` is a DataFrame or
        a dict of per-env columns (tensors are copied to the host with everything else)."""
        if extra is not None and len(extra) != self.parallel_envs and not isinstance(extra, dict):
            raise ValueError('The number of elements in extras must match the number of parallel environments.')
        self._raise_failure()
        agent_tuple = tuple(agents)
        if self._agent_set is None or reset:
            self._agent_set = agent_tuple
        elif agent_tuple != self._agent_set:
            raise RuntimeError(f'CSVLogger does not support changing agents mid-simulation. '
                               f'Initial agents: {self._agent_set}, current agents: {agent_tuple}.')

        job: Dict[str, Any] = {'kind': 'row', 'reset': reset, 'agents': agent_tuple, 'description': log_description}
        job['state'] = state.to_dataframe_parts(_to_host)
        if not reset:
            job['actions'] = {agent: _to_host(actions[agent]) for agent in agents}
            job['rewards'] = {agent: _to_host(rewards[agent]) for agent in agents}
            job['num_moves'], job['finished'] = _to_host(num_moves), _to_host(finished)
        job['action_map'] = {agent: _snapshot_mapping(agent_action_mapping[agent]) for agent in agents}
        job['observation_map'] = {agent: _snapshot_mapping(agent_observation_mapping[agent]) for agent in agents}
        if isinstance(extra, dict):
            extra = {key: _to_host(value) for key, value in extra.items()}
        job['extra'] = extra
        job['event'] = self._completion_event()
        self._submit(job)

    def reset(self, *args, **kwargs):
        self._submit({'kind': 'reset'})

    # --------------------------------------------------------------------------------------------- writer side
    def _handle(self, job) -> None:
        if job['kind'] == 'reset':
            self._are_logs_initialized = False
            return
        if job['event'] is not None:
            job['event'].synchronize()
        reset, agents = job['reset'], job['agents']
        if reset:
            self._are_logs_initialized = False

        columns, shared = job['state']
        df = pd.DataFrame({name: [str(row.tolist()) for row in value] for name, value in columns})
        for name, value in shared:
            df[name] = str(value.tolist())
        rows = len(df)
        new_cols = {}
        if reset:
            for agent in agents:
                new_cols[f'{agent}_action'] = [None] * rows
                new_cols[f'{agent}_rewards'] = [None] * rows
            new_cols['step'] = [-1] * rows
            new_cols['complete'] = [None] * rows
        else:
            for agent in agents:
                new_cols[f'{agent}_action'] = [str(action) for action in job['actions'][agent].tolist()]
                new_cols[f'{agent}_rewards'] = job['rewards'][agent].tolist()
            new_cols['step'] = job['num_moves'].tolist()
            new_cols['complete'] = job['finished'].tolist()
        for agent in agents:
            new_cols[f'{agent}_action_map'] = _mapping_rows(job['action_map'][agent])
            new_cols[f'{agent}_observation_map'] = _mapping_rows(job['observation_map'][agent])
        df = pd.concat([df, pd.DataFrame(new_cols)], axis=1)

        extra = job['extra']
        if extra is not None:
            if isinstance(extra, dict):
                extra = pd.DataFrame({key: (value.tolist() if isinstance(value, torch.Tensor) else value) for key, value in extra.items()})
            df = pd.concat([df, extra], axis=1)

        df['description'] = job['description']
        for i in range(self.parallel_envs):
            df.iloc[[i]].to_csv(
                os.path.join(self.log_directory, f'{i}.csv'),
                mode='w' if not self._are_logs_initialized else 'a',
                header=not self._are_logs_initialized,
                index=False,
                na_rep='NULL',
            )
        self._are_logs_initialized = True


class SQLLogger(_TapLogger):
    """SQL logger (logging_handlers.py:117-241) over the schema of utils/sql_logging.py:12-118 — tables ``simulation``, ``environment``,
    ``agent``, ``environment_timestep``, ``<domain>_environment_log``, ``agent_log`` with the reference's column names, so that a
    database written here reads like one written by the reference (its converter included).  ``reset`` opens a simulation with one
    environment row per parallel env and one agent row per (agent, env); every logged step adds, per env, a timestep row, the
    domain's state row (cells are ``str(tensor.tolist())``) and — except on a reset row — one ``agent_log`` row per agent
    (reward truncated to an integer, as the reference stores it).  Rows are inserted table by table in the reference's per-table
    order, one transaction per step, on the writer thread."""

    _thread_name = 'frz-sql-logger'
    _STATE_COLUMNS = {
        'wildfire': (('fires', 'intensity', 'fuel', 'suppressants', 'capacity', 'equipment'), ('agents', )),
        'rideshare': (('agents', 'passengers'), ()),
        'cybersecurity': (('network_state', 'location', 'presence'), ()),
    }

    def __init__(self, connection_string: str, domain: str, parallel_envs: int, asynchronous: bool = True):
        super().__init__(asynchronous)
        import sqlalchemy as sa
        self.connection_string, self.domain, self.parallel_envs = connection_string, domain, parallel_envs
        self._kind = domain.split('_')[0]
        if self._kind not in self._STATE_COLUMNS:
            raise NotImplementedError(f'Environment {domain} does not have an implemented log_environment function.')
        self._sa = sa
        meta = sa.MetaData()
        ident = lambda: sa.Column('id', sa.Integer, primary_key=True)  # noqa: E731
        fk = lambda name, target: sa.Column(name, sa.Integer, sa.ForeignKey(target), nullable=False)  # noqa: E731
        text = lambda *names: [sa.Column(name, sa.Text) for name in names]  # noqa: E731
        self._simulation = sa.Table('simulation', meta, ident(), sa.Column('name', sa.Text, nullable=False), sa.Column('description', sa.Text),
                                    sa.Column('timestamp', sa.Date, nullable=False))
        self._environment = sa.Table('environment', meta, ident(), fk('simulation_id', 'simulation.id'), sa.Column('simulation_index', sa.Integer))
        self._agent = sa.Table('agent', meta, ident(), fk('environment_id', 'environment.id'), sa.Column('name', sa.Text, nullable=False))
        self._timestep = sa.Table('environment_timestep', meta, fk('environment_id', 'environment.id'), ident(), sa.Column('timestep', sa.Integer))
        self._state_tables = {
            'wildfire': sa.Table('wildfire_environment_log', meta, ident(), fk('simulation_timestep_id', 'environment_timestep.id'),
                                 *text('fires', 'intensity', 'fuel', 'suppressants', 'capacity', 'equipment', 'agents')),
            'rideshare': sa.Table('rideshare_environment_log', meta, ident(), fk('simulation_timestep_id', 'environment_timestep.id'),
                                  *text('agents', 'passengers')),
            'cybersecurity': sa.Table('cybersecurity_environment_log', meta, ident(), fk('simulation_timestep_id', 'environment_timestep.id'),
                                      *text('network_state', 'location', 'presence', 'adj_matrix')),
        }
        self._agent_log = sa.Table('agent_log', meta, ident(), fk('simulation_timestep_id', 'environment_timestep.id'), fk('agent_id', 'agent.id'),
                                   sa.Column('reward', sa.Integer), sa.Column('action_field', sa.Integer), sa.Column('task_field', sa.Integer),
                                   *text('action_map', 'observation_map'))
        self.engine = sa.create_engine(connection_string)
        meta.create_all(self.engine)
        self._env_ids: Optional[List[int]] = None
        self._agent_ids: Dict[Any, int] = {}
        self._opened = False

    # ------------------------------------------------------------------------------------------- producer side
    def reset(self, log_label=None, log_description=None, agents=None):
        self._raise_failure()
        self._opened = True
        self._submit({'kind': 'reset', 'label': log_label, 'description': log_description, 'agents': None if agents is None else tuple(agents)})

    def log_environment(self,
                        state,
                        actions,
                        rewards,
                        agent_action_mapping,
                        agent_observation_mapping,
                        num_moves,
                        finished,
                        log_description,
                        agents,
                        extra=None,
                        reset=False):
        if not self._opened:
            raise RuntimeError('SQLLogger: reset() must be called before logging. _env_ids is None.')
        self._raise_failure()
        per_env, shared = self._STATE_COLUMNS[self._kind]
        job: Dict[str, Any] = {'kind': 'row', 'reset': reset, 'agents': tuple(agents)}
        job['state'] = {name: _to_host(getattr(state, name)) for name in per_env + shared}
        job['num_moves'] = _to_host(num_moves)
        if not reset:
            job['actions'] = {agent: _to_host(actions[agent]) for agent in agents}
            job['rewards'] = {agent: _to_host(rewards[agent]) for agent in agents}
            job['action_map'] = {agent: _snapshot_mapping(agent_action_mapping[agent]) for agent in agents}
            job['observation_map'] = {agent: _snapshot_mapping(agent_observation_mapping[agent]) for agent in agents}
        job['extra'] = extra
        job['event'] = self._completion_event()
        self._submit(job)

    # --------------------------------------------------------------------------------------------- writer side
    def _insert(self, connection, table, rows) -> List[int]:
        """Insert ``rows`` in order; the generated ids in the same order."""
        if not rows:
            return []
        statement = self._sa.insert(table).returning(table.c.id, sort_by_parameter_order=True)
        return [row[0] for row in connection.execute(statement, rows)]

    def _handle(self, job) -> None:
        if job['kind'] == 'reset':
            with self.engine.begin() as connection:
                simulation, = self._insert(connection, self._simulation, [{'name': job['label'] or 'simulation', 'description': job['description'],
                                                                          'timestamp': datetime.datetime.now().date()}])
                self._env_ids = self._insert(connection, self._environment,
                                             [{'simulation_id': simulation, 'simulation_index': i} for i in range(self.parallel_envs)])
                self._agent_ids = {}
                if job['agents'] is not None:
                    keys = [(agent, env_id) for agent in job['agents'] for env_id in self._env_ids]
                    ids = self._insert(connection, self._agent, [{'name': agent, 'environment_id': env_id} for agent, env_id in keys])
                    self._agent_ids = dict(zip(keys, ids))
            return
        if job['event'] is not None:
            job['event'].synchronize()
        per_env, shared = self._STATE_COLUMNS[self._kind]
        state, agents, envs = job['state'], job['agents'], range(self.parallel_envs)
        moves = job['num_moves'].tolist()
        with self.engine.begin() as connection:
            timesteps = self._insert(connection, self._timestep, [{'environment_id': self._env_ids[i], 'timestep': int(moves[i])} for i in envs])
            rows = []
            for i in envs:
                row = {'simulation_timestep_id': timesteps[i]}
                row.update({name: str(state[name][i].tolist()) for name in per_env})
                row.update({name: str(state[name].tolist()) for name in shared})
                if self._kind == 'cybersecurity':
                    row['adj_matrix'] = str(job['extra']['adj_matrix'][i])
                rows.append(row)
            self._insert(connection, self._state_tables[self._kind], rows)
            # agents that were not among the possible agents at reset() join the simulation here (logging_handlers.py:215-221)
            missing = [(agent, self._env_ids[i]) for i in envs for agent in agents if (agent, self._env_ids[i]) not in self._agent_ids]
            if missing:
                ids = self._insert(connection, self._agent, [{'name': agent, 'environment_id': env_id} for agent, env_id in missing])
                self._agent_ids.update(zip(missing, ids))
            if not job['reset']:
                actions = {agent: job['actions'][agent].tolist() for agent in agents}
                rewards = {agent: job['rewards'][agent].tolist() for agent in agents}
                action_maps = {agent: _mapping_rows(job['action_map'][agent]) for agent in agents}
                observation_maps = {agent: _mapping_rows(job['observation_map'][agent]) for agent in agents}
                self._insert(connection, self._agent_log, [{
                    'simulation_timestep_id': timesteps[i], 'agent_id': self._agent_ids[(agent, self._env_ids[i])],
                    'reward': int(rewards[agent][i]), 'action_field': int(actions[agent][i][1]), 'task_field': int(actions[agent][i][0]),
                    'action_map': action_maps[agent][i], 'observation_map': observation_maps[agent][i]} for i in envs for agent in agents])
