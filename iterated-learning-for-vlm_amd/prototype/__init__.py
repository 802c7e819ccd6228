"""Host-side mirror of the reference `prototype` package for the contrastive hot path: same module
paths, registry strings, class and parameter names; arithmetic runs on the HIP kernels of libilvlm_hip.so."""
