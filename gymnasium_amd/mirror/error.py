"""Exception types of the path (mirror of gymnasium/error.py; same names, same base classes)."""


class Error(Exception):
    """Base error of the package (gymnasium.error.Error)."""


class UnregisteredEnv(Error):
    """Unknown environment id."""


class NamespaceNotFound(UnregisteredEnv):
    """Unknown namespace."""


class NameNotFound(UnregisteredEnv):
    """Unknown name."""


class VersionNotFound(UnregisteredEnv):
    """Unknown version."""


class RegistrationError(Error):
    """Bad ``register`` arguments."""


class DependencyNotInstalled(Error):
    """A required dependency is missing."""


class ResetNeeded(Error):
    """``step`` called before ``reset``."""


class InvalidAction(Error):
    """Action outside the action space."""


class ClosedEnvironmentError(Error):
    """Use after ``close``."""


class InvalidBound(Error):
    """Raised when the clipping an array with invalid upper and/or lower bound (gymnasium/error.py:62-63)."""


class AlreadyPendingCallError(Exception):
    """gymnasium/error.py:72-80: an asynchronous call (`step_async`) is outstanding and another call is made before its `step_wait`."""

    def __init__(self, message: str, name: str):
        super().__init__(message)
        self.name = name


class NoAsyncCallError(Exception):
    """gymnasium/error.py:83-91: `step_wait` without a preceding `step_async`."""

    def __init__(self, message: str, name: str):
        super().__init__(message)
        self.name = name
