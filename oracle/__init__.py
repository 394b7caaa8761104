"""CPU oracle (test infrastructure).  See oracle/igw_oracle.h."""
