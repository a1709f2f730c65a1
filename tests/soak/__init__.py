"""Long-horizon parity runs against the oracle (run by hand on the GPU box; results under profiles/).  They live under tests/ because
they use the oracle, which is test infrastructure: nothing outside tests/, smoke() and bench.py's cpu_baseline leg may."""
