from .JESMOC_MFDGP import JESMOC_MFDGP, _JES_MFDGP, optimize_acqf_multistart
