"""Measurement, roofline and reporting helpers of bench.py (the driver contract lives in bench.py itself)."""
