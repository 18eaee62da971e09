"""TEST INFRASTRUCTURE ONLY.

CPU oracle for the KKT factor/solve hot path (see oracle/kvx_oracle.c header).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; nothing under kvxopt_amd/ does.
"""
