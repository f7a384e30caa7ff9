"""CPU oracle for the QCMRF statevector hot path.  TEST INFRASTRUCTURE ONLY.

Nothing in the shipped package (``qcmrf_amd/``) may import, call, link or execute
anything in this directory.  Allowed users: ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` -- and there only as the checker /
the thing timed beside the GPU path, never as the product.

Contents
--------
closed_form.py   exact output distribution / amplitudes of a QCMRF circuit, derived
                 line by line from /root/reference/QCMRF.py:199-243.
gate_stream.py   independent restatement of the reference gate stream
                 (QCMRF.py:199-243 + Qiskit's documented AND / cp / h / x semantics).
sv_numpy.py      gate-level fp64 numpy statevector simulator (small W).
qsv_ref.c        the same gate-level simulator in plain C + OpenMP (W up to ~30 on a
                 big host); built by oracle/Makefile into oracle/_build/libqsv_ref.so.
cref.py          ctypes binding of qsv_ref.c.

Parity pin status
-----------------
The path's arithmetic lives in Qiskit Aer (third party, NOT vendored in the reference,
version unpinned -- imports imply qiskit-terra 0.45/0.46 + qiskit-aer 0.13.x).  Neither
is installed here, so the reference cannot be run.  The oracle is pinned by the
reference's own committed Aer outputs ``res_{0.1,0.25,0.5}/result_simulation.json``
(210 circuits x 10 000 shots; tests/test_golden_aer.py): chi^2/dof ~= 1.0 for the
conventions used here, 40..1400 for any wrong bit/ancilla/theta convention.
That pins the *distribution and every convention* at the ~1e-2 statistical level.
At the 1e-10 level **parity is unpinned by the reference** (it commits no amplitude or
probability vector); it is closed by exact algebra (closed form == gate-level
simulation to ~1e-15) -- see DESIGN.md.
"""
