"""TEST INFRASTRUCTURE ONLY -- float64 CPU restatement of the reference hot path.

Nothing under ``multidronesim_amd/`` may import this package.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use it, and
only as the checker / the timed CPU baseline -- never as the product path.
"""
