"""oracle/ -- TEST INFRASTRUCTURE ONLY.  CPU checkers for the SpMV hot path:

* ``liboracle.so``           plain-C restatement of the reference's host arithmetic (spmv_oracle.c)
* ``_ref/libcusp_ref.so``    the reference's OWN sequential kernels, compiled from /root/reference
                             (ref_shim.cpp; built only where the reference tree exists)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package, and
only as the checker.  The product (cusp-autotuned_amd/) never does.
"""
from .loader import Oracle, Reference, build, have_reference, fill_x  # noqa: F401
