"""``from mobocmf_amd.models.mfgp_lin import MFGP_lin`` -- same module path as the reference's mobocmf/models/mfgp_lin.py."""
from .mfgp import MFGP_lin, MFKernel_lin, TL  # noqa: F401
