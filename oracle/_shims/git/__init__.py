"""Stub of GitPython for importing the reference. Test infrastructure only."""


class InvalidGitRepositoryError(Exception):
    pass


class Repo:
    def __init__(self, *a, **k):
        raise InvalidGitRepositoryError()
