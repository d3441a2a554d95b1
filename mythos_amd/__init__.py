"""mythos_amd: MI355X-native (HIP / gfx950) force, parameter-gradient and Langevin core for the oxDNA and MARTINI
energy functions of mythos-bio/mythos, behind a C ABI (include/mythos_hip.h) and a torch-facing Python surface that
mirrors the reference's energy-function / simulator / DiffTRe protocols.  No CPU fallback: see DESIGN.md."""

__version__ = "0.1.0"
