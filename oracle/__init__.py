"""CPU oracle for the MPPI hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  The product package (ccv_mppi_path_tracker_amd) never does.
Parity unpinned: see the header of mppi_oracle.cpp and DESIGN.md.
"""
