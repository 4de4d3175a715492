"""strelka_amd -- MI355X-native implementation of the Strelka2 per-region hot path (candidate-alignment scoring and
per-locus genotype likelihoods) behind the C-ABI of include/strelka_amd.h.

Python here is plumbing (ctypes binding, synthetic inputs, build driver); the product is strelka_amd/lib/libstrelka_amd.so.
"""
__all__ = ["capi", "synth", "build"]
