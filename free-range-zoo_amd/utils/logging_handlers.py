"""Logging tap for the batched environments (mirrors free_range_zoo/utils/logging_handlers.py:21-114).

Same interface and the same files as the reference's ``CSVLogger`` — one ``<env>.csv`` per parallel environment, one row per
reset / step, identical columns and cell text — but taken off the step path: ``log_environment`` only enqueues device-to-host
copies of what a row needs on the env's stream (pinned destinations, no host synchronisation) followed by an event; a writer thread
waits for the event, formats the rows with pandas exactly as the reference does and appends them.  The step loop never waits for
the file system; ``flush()`` / ``close()`` drain the queue.

The SQL logger of the reference (``SQLLogger``, logging_handlers.py:117-241, sqlite / postgres through SQLAlchemy) is not built.
"""
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


class CSVLogger(Logger):
    """CSV logger writing the reference's format (logging_handlers.py:36-114) from a background writer thread."""

    def __init__(self, log_directory: str, parallel_envs: int, override_initialization_check: bool = False, asynchronous: bool = True):
        self.log_directory = log_directory
        self.parallel_envs = parallel_envs
        self._are_logs_initialized = False
        self._agent_set = None
        if not override_initialization_check and os.path.exists(log_directory):
            if os.listdir(log_directory):
                raise FileExistsError('The logging output directory already exists. Set override_initialization_check or rename.')
        if not os.path.exists(log_directory):
            os.mkdir(log_directory)
        self._asynchronous = asynchronous
        self._jobs: 'queue.Queue' = queue.Queue()
        self._failure: Optional[BaseException] = None
        self._worker: Optional[threading.Thread] = None

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
        job['event'] = None
        if torch.cuda.is_available() and torch.cuda.is_initialized():
            job['event'] = torch.cuda.Event()
            job['event'].record()
        self._submit(job)

    def reset(self, *args, **kwargs):
        self._submit({'kind': 'reset'})

    def flush(self) -> None:
        """Block until every row handed over so far is in its file."""
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
            raise RuntimeError('the CSV writer thread failed') from failure

    def _submit(self, job) -> None:
        if not self._asynchronous:
            self._handle(job)
            return
        if self._worker is None:
            self._worker = threading.Thread(target=self._drain, name='frz-csv-logger', daemon=True)
            self._worker.start()
        self._jobs.put(job)

    # --------------------------------------------------------------------------------------------- writer side
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
