"""Clustering (reference: python/cuvs/cuvs/cluster/)."""
from . import kmeans  # noqa: F401
