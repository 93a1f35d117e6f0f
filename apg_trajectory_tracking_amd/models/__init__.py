"""Policy networks of the APG hot path (PyTorch-ROCm; rocBLAS/MIOpen do the
work - these stay tiny MLP / conv1d / LSTMCell modules as in the reference).
Parameter names match the reference modules, so a reference `state_dict()`
loads unchanged."""
