"""CPU oracle (test infrastructure only — see apla_oracle.py header)."""
