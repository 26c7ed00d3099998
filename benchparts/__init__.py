"""Parts of bench.py (repo root): one module per leg of the benchmark."""
