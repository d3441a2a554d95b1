"""Generators of synthetic systems (ideal B-duplexes, bundles) used by the benchmarks and the size tests."""
