"""Energy functions with the reference's surface (oxDNA1, oxDNA2, MARTINI 2/3) evaluated by the HIP kernels."""
