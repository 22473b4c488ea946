"""Pieces of bench.py that need neither the oracle nor a workload: rooflines / PMC sidecars (roofline.py) and the process-group,
timing and frame-parallel self-check plumbing (distrib.py).  bench.py at the repository root stays the entry point the driver runs."""
