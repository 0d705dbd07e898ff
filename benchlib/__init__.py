"""Support code of bench.py (the driver-facing script at the repo root): one module per leg."""
