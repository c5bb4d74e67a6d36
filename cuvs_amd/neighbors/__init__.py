from . import brute_force, cagra, ivf_flat, ivf_pq, mg  # noqa: F401
from .refine import refine  # noqa: F401
