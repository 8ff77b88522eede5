"""TEST INFRASTRUCTURE -- the CPU oracle.  Importable only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
