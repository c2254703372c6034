"""Driver-level drop-ins: an ex2b-compatible command line and a
figure_gen-compatible experiment runner, both on top of the device solvers."""
