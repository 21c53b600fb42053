"""Stub of `loguru` so the (unmodified) reference imports in this container. Test infrastructure only."""


class _Quiet:
    def __getattr__(self, name):
        return lambda *a, **k: None


logger = _Quiet()
