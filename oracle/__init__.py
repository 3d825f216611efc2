"""CPU oracle for the CloserLook3D local-aggregation hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package.  The product package ``closerlook3d_amd`` never does (tests/test_abi.py greps for it).

* ``oracle.native``     ctypes binding of ``cl3d_oracle.c`` (restatement of the five ``.cu`` kernels)
* ``oracle.operators``  torch-CPU fp32 restatement of pt_utils.py / local_aggregation_operators.py
* ``oracle.build_ref``  recipe that compiles the reference's own extension for gfx950 into
                        ``oracle/_ref`` (only runs where /root/reference exists)
"""
