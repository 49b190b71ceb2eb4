"""CPU oracle for the next-plaid search path -- TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only.
See plaid_oracle.c for the parity-pinning statement.
"""
