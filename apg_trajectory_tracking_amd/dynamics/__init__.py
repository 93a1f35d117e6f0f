"""Mirror of `neural_control.dynamics` for the APG hot path."""
