from . import brute_force  # noqa: F401
