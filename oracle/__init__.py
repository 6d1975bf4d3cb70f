"""TEST INFRASTRUCTURE ONLY -- float64 CPU restatements of the reference hot path.

  np_oracle.py        NumPy restatement of every piece on the path (pinned by tests/golden/*.npz, minted from the reference)
  np_trajectories.py  the other trajectory classes
  c_oracle.c / .py    a second, separately written restatement of the C2 / C3 loop in plain C (make -C oracle); also the timed CPU baseline
  cvxopt_qp.py        the reference's third-party QP solver (cvxopt 1.3.2 coneqp) restated from its published algorithm -- UNPINNED

Nothing under ``multidronesim_amd/`` may import this package.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use it, and
only as the checker / the timed CPU baseline -- never as the product path.
"""
