"""CPU restatements of the reference arithmetic - TEST INFRASTRUCTURE ONLY (never imported by mythos_amd)."""
