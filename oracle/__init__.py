"""CPU oracle (test infrastructure only) -- see oracle/mld_oracle.py for scope and pinning status."""
