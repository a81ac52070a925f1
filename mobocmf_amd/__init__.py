"""mobocmf_amd: MI355X-native variational multi-fidelity deep-GP (MFDGP) layer + ELBO hot path.

Mirrors the reference surface ``mobocmf.models.MFDGP`` / ``mobocmf.mlls.VariationalELBOMF`` (see
DESIGN.md); all arithmetic runs in hand-written HIP kernels behind the C-ABI of include/mobocmf_hip.h.
"""
__version__ = "0.1.0"
