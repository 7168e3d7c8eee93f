"""CPU oracle for the Kuro Siwo training hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it, and only as the *checker* (never as the thing measured as the
GPU path or shipped).  The product package ``kurosiwo_amd`` never imports it.

The oracle is a functional (state-dict driven) fp32 restatement, on stock
PyTorch-CPU ops, of what the reference's ``nn.Module`` graphs compute.  Each
function cites the reference ``file:line`` it follows.  It is pinned to the
real reference by the golden vectors in ``tests/golden/`` which were produced
by importing ``/root/reference`` in the build container
(``oracle/gen_golden.py``; the reference itself never travels).
"""
