from .mfdgp_hidden_layer import (MFDGPHiddenLayer, MFDGUnwhitenedVariationalStrategy, NotPSDError,
                                 UnwhitenedVariationalStrategy)
