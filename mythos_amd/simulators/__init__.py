"""Simulator protocol, trajectory container, neighbour helpers and the HIP MD simulator (fused Langevin kernel)."""
