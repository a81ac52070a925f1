# Empty, as in the reference: ``from ...acquisition_functions.JESMOC_MFDGP import JESMOC_MFDGP``.  Re-exporting the class
# here would shadow the submodule of the same name and make its classes unlocatable for dill (the examples pickle the
# acquisition object, example_acquisition_mfdgp_forrester.py:140-142).
