"""Host runtime under the reference's Python op surface: ctypes binding of libtg_hip.so,
geometry descriptors, persistent device workspace, tape, RNG, hipGraph and DP helpers."""
