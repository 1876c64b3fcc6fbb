"""TEST INFRASTRUCTURE ONLY.

CPU restatement (oracle) of the CSTP R(2+1)D-BYOL pre-training step.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this package; the product path (``cstp_amd``) never does.
"""
