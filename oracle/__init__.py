"""CPU oracle package -- test infrastructure only (see w2l_oracle.py header)."""
