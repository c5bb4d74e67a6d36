from . import brute_force, ivf_pq  # noqa: F401
