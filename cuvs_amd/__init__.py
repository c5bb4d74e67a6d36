"""cuvs_amd — MI355X-native vector-search hot path behind the cuvs C ABI.

Python surface mirrors python/cuvs of the reference (cuvs.neighbors.{brute_force,ivf_flat,ivf_pq,cagra},
cuvs.cluster.kmeans, cuvs.common.Resources) so tests read like the reference's own tests.
"""
from . import cluster, common, distance, neighbors  # noqa: F401

__version__ = "26.08.00"
