"""Parity tests: `-m "not gpu"` (oracle vs golden files, host logic, ABI symbols) and `-m gpu` (HIP path vs oracle)."""
