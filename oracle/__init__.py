"""TEST INFRASTRUCTURE ONLY: parity oracle (see oracle/pb_oracle.h)."""
