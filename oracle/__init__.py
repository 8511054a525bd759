"""ORACLE -- test infrastructure only; parity unpinned (see oracle/oracle.py)."""
