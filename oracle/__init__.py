"""CPU oracle for msmv_sampling.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package; the product package ``opus_amd`` never does.
"""
