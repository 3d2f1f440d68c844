"""Device engines: orchestration of libinfernos_hip.so kernels for each model family."""
