"""simulations/ of the reference, restricted to the callers of the hot path:
``EnvGeometric.GeometricEnv.do_control`` (configs 2/3), ``CBFTest`` (config 4), ``CBFTestOrd3``.
The FedCE / dLQR system-identification loops of those scripts stay with the reference."""
