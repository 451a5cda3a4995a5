"""Minimal stand-in for `astropy`, used ONLY by oracle/make_golden.py (test infrastructure)."""
