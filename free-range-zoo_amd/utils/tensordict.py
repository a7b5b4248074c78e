"""``TensorDict`` as the reference uses it for observations (``TensorDict({'self','others','tasks'}, batch_size=[B])``).

The real ``tensordict`` package is used when it is installed; otherwise a dict subclass with the two attributes the
reference's callers touch (``batch_size``, ``device``)."""
try:  # pragma: no cover - depends on the image
    from tensordict import TensorDict  # type: ignore
except Exception:  # noqa: BLE001

    class TensorDict(dict):
        def __init__(self, source=None, batch_size=None, device=None, **kwargs):
            super().__init__(source or {})
            self.batch_size = list(batch_size) if batch_size is not None else []
            self.device = device

        def to(self, device):
            return TensorDict({k: v.to(device) for k, v in self.items()}, batch_size=self.batch_size, device=device)
