"""warn()/error() through the warnings module (mirror of gymnasium/logger.py:17-47)."""
import warnings

DEBUG, INFO, WARN, ERROR, DISABLED = 10, 20, 30, 40, 50
min_level = 30
warnings.filterwarnings("once", "", DeprecationWarning, module=r"^gymnasium_amd\.")


def warn(msg, *args, category=None, stacklevel=1):
    if min_level <= WARN:
        warnings.warn(f"\x1b[33mWARN: {msg % args}\x1b[0m", category=category, stacklevel=stacklevel + 1)


def deprecation(msg, *args):
    warn(msg, *args, category=DeprecationWarning, stacklevel=2)


def error(msg, *args):
    if min_level <= ERROR:
        warnings.warn(f"\x1b[31mERROR: {msg % args}\x1b[0m", stacklevel=3)
