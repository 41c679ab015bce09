"""CPU oracle for the DSI hot path -- TEST INFRASTRUCTURE ONLY (parity unpinned).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  See oracle/dsi_oracle.h for what it restates and why it is
unpinned.
"""
