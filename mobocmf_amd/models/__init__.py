from .mfdgp import MFDGP, TL
